"""Oracle (TEST INFRASTRUCTURE ONLY): U2-Net-P saliency + SAMP-Net composition in torch-CPU fp32.

Restates reference models/samp_net.py: REBNCONV :45-54, RSU7/6/5/4/4F :62-255, U2NETP :258-342,
SAMPPModule :429-645, SAMPNet :665-791 and the scorer post-processing :957-989. State-dict keys are the
reference's, so one checkpoint loads into both. PINNED: tests/test_oracle_golden.py checks this file
against golden vectors produced by the reference's own classes (tests/golden/make_samp_golden.py).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .resnet import resnet18_trunk

COMPOSITION_PATTERNS = ['global', 'horizontal', 'vertical', 'triangular', 'surround', 'quarter', 'cross',
                        'rule_of_thirds']  # samp_net.py:23-32


class _CBR(nn.Module):  # samp_net.py:45-54
    def __init__(self, cin, cout, d):
        super().__init__()
        self.conv_s1 = nn.Conv2d(cin, cout, 3, padding=d, dilation=d)
        self.bn_s1 = nn.BatchNorm2d(cout)

    def forward(self, x):
        return F.relu(self.bn_s1(self.conv_s1(x)))


def _up(src, like):  # samp_net.py:57-59
    return F.interpolate(src, size=like.shape[2:], mode='bilinear', align_corners=False)


class _RSU(nn.Module):
    """RSU-L with pooling (samp_net.py:62-229): encoder L-1 levels, dilated bottom, decoder with skip concat."""

    def __init__(self, depth, cin, mid, cout):
        super().__init__()
        self.depth = depth
        self.rebnconvin = _CBR(cin, cout, 1)
        self.rebnconv1 = _CBR(cout, mid, 1)
        for k in range(2, depth):
            setattr(self, f"rebnconv{k}", _CBR(mid, mid, 1))
        setattr(self, f"rebnconv{depth}", _CBR(mid, mid, 2))
        for k in range(depth - 1, 1, -1):
            setattr(self, f"rebnconv{k}d", _CBR(mid * 2, mid, 1))
        self.rebnconv1d = _CBR(mid * 2, cout, 1)

    def forward(self, x):
        L = self.depth
        hin = self.rebnconvin(x)
        enc = [self.rebnconv1(hin)]
        for k in range(2, L):
            enc.append(getattr(self, f"rebnconv{k}")(F.max_pool2d(enc[-1], 2, 2, ceil_mode=True)))
        bottom = getattr(self, f"rebnconv{L}")(enc[-1])
        d = getattr(self, f"rebnconv{L - 1}d")(torch.cat((bottom, enc[-1]), 1)) if L > 2 else None
        for k in range(L - 2, 0, -1):
            d = getattr(self, f"rebnconv{k}d")(torch.cat((_up(d, enc[k - 1]), enc[k - 1]), 1))
        return d + hin


class _RSU4F(nn.Module):  # samp_net.py:232-255
    def __init__(self, cin, mid, cout):
        super().__init__()
        self.rebnconvin = _CBR(cin, cout, 1)
        self.rebnconv1 = _CBR(cout, mid, 1)
        self.rebnconv2 = _CBR(mid, mid, 2)
        self.rebnconv3 = _CBR(mid, mid, 4)
        self.rebnconv4 = _CBR(mid, mid, 8)
        self.rebnconv3d = _CBR(mid * 2, mid, 4)
        self.rebnconv2d = _CBR(mid * 2, mid, 2)
        self.rebnconv1d = _CBR(mid * 2, cout, 1)

    def forward(self, x):
        hin = self.rebnconvin(x)
        h1 = self.rebnconv1(hin)
        h2 = self.rebnconv2(h1)
        h3 = self.rebnconv3(h2)
        h4 = self.rebnconv4(h3)
        h3d = self.rebnconv3d(torch.cat((h4, h3), 1))
        h2d = self.rebnconv2d(torch.cat((h3d, h2), 1))
        h1d = self.rebnconv1d(torch.cat((h2d, h1), 1))
        return h1d + hin


class U2NETP(nn.Module):  # samp_net.py:258-342
    def __init__(self):
        super().__init__()
        self.stage1 = _RSU(7, 3, 16, 64)
        self.stage2 = _RSU(6, 64, 16, 64)
        self.stage3 = _RSU(5, 64, 16, 64)
        self.stage4 = _RSU(4, 64, 16, 64)
        self.stage5 = _RSU4F(64, 16, 64)
        self.stage6 = _RSU4F(64, 16, 64)
        self.stage5d = _RSU4F(128, 16, 64)
        self.stage4d = _RSU(4, 128, 16, 64)
        self.stage3d = _RSU(5, 128, 16, 64)
        self.stage2d = _RSU(6, 128, 16, 64)
        self.stage1d = _RSU(7, 128, 16, 64)
        for k in range(1, 7):
            setattr(self, f"side{k}", nn.Conv2d(64, 1, 3, padding=1))
        self.outconv = nn.Conv2d(6, 1, 1)

    def forward(self, x):
        pool = lambda t: F.max_pool2d(t, 2, 2, ceil_mode=True)
        h1 = self.stage1(x)
        h2 = self.stage2(pool(h1))
        h3 = self.stage3(pool(h2))
        h4 = self.stage4(pool(h3))
        h5 = self.stage5(pool(h4))
        h6 = self.stage6(pool(h5))
        h5d = self.stage5d(torch.cat((_up(h6, h5), h5), 1))
        h4d = self.stage4d(torch.cat((_up(h5d, h4), h4), 1))
        h3d = self.stage3d(torch.cat((_up(h4d, h3), h3), 1))
        h2d = self.stage2d(torch.cat((_up(h3d, h2), h2), 1))
        h1d = self.stage1d(torch.cat((_up(h2d, h1), h1), 1))
        d1 = self.side1(h1d)
        sides = [d1] + [_up(getattr(self, f"side{k}")(t), d1)
                        for k, t in zip(range(2, 7), (h2d, h3d, h4d, h5d, h6))]
        d0 = self.outconv(torch.cat(sides, 1))
        return torch.sigmoid(d0)  # SaliencyDetector.detect uses d0 only (samp_net.py:421-422)


PATTERN_SHAPES = [(1296, 2, 1), (1296, 1, 2), (1373, 2, 1), (1373, 2, 1), (1296, 2, 1), (1296, 2, 2), (1324, 2, 2),
                  (836, 3, 3)]


def _gmax(t):
    return F.adaptive_max_pool2d(t, 1).flatten(1)


def _gavg(t):
    return F.adaptive_avg_pool2d(t, 1).flatten(1)


def regional_features(fm, sal, idx):
    """samp_net.py:463-596: the 8 region-pool feature vectors, padded with tiled global_max or truncated to
    the checkpoint's conv input size and *reshaped* (not permuted) to [B, C, kh, kw]."""
    B, C, H, W = fm.shape
    gmax, gavg = _gmax(fm), _gavg(fm)
    sal_small = F.adaptive_avg_pool2d(sal, (4, 4)).flatten(1)
    top, bot = fm[:, :, :H // 2], fm[:, :, H // 2:]
    ctr = fm[:, :, H // 4:3 * H // 4, W // 4:3 * W // 4]
    if idx in (0, 1, 4):
        if idx == 0:
            r1, r2 = top, bot
        elif idx == 1:
            r1, r2 = fm[..., :W // 2], fm[..., W // 2:]
        else:
            r1, r2 = fm[:, :, H // 4:H - H // 4, W // 4:W - W // 4], fm
        parts = [_gmax(r1), _gavg(r1), _gmax(r2), _gavg(r2), sal_small]
    elif idx in (2, 3):
        parts = [_gmax(top), _gavg(top), _gmax(bot), _gavg(bot), _gmax(ctr), sal_small]
    elif idx == 5:
        qs = [fm[:, :, :H // 2, :W // 2], fm[:, :, :H // 2, W // 2:], fm[:, :, H // 2:, :W // 2],
              fm[:, :, H // 2:, W // 2:]]
        parts = [_gmax(q) for q in qs] + [_gavg(q) for q in qs] + [_gmax(ctr), sal_small]
    elif idx == 6:
        h3, w3 = H // 3, W // 3
        parts = [_gmax(fm[:, :, i * h3:min(H, (i + 1) * h3), j * w3:min(W, (j + 1) * w3)])
                 for i in range(3) for j in range(3)] + [sal_small]
    else:  # idx == 7
        parts = [gmax, gavg] + [F.adaptive_avg_pool2d(fm, s).flatten(1) for s in (2, 3, 4)]
        sw = F.interpolate(sal, size=(H, W), mode='bilinear', align_corners=False)
        parts += [_gavg(fm * sw), F.adaptive_avg_pool2d(sal, (8, 8)).flatten(1)]
    feat = torch.cat(parts, 1)
    c, kh, kw = PATTERN_SHAPES[idx]
    need = c * kh * kw
    if feat.shape[1] < need:
        pad = need - feat.shape[1]
        feat = torch.cat([feat, gmax.repeat(1, pad // C + 1)[:, :pad]], 1)
    else:
        feat = feat[:, :need]
    return feat.view(B, c, kh, kw)


class _Patterns(nn.Module):  # SAMPPModule, samp_net.py:429-645
    def __init__(self):
        super().__init__()
        self.conv_list = nn.ModuleList(
            [nn.Sequential(nn.Conv2d(c, 1024, (kh, kw), bias=False)) for c, kh, kw in PATTERN_SHAPES])

    def forward(self, fm, sal_down, pw):
        sal = F.interpolate(sal_down, size=fm.shape[2:], mode='bilinear', align_corners=False)
        feats = torch.stack([self.conv_list[i](regional_features(fm, sal, i)).flatten(1) for i in range(8)], 1)
        return (feats * F.softmax(pw, 1).unsqueeze(2)).sum(1)


class SAMPNet(nn.Module):  # samp_net.py:665-791 (eval mode: dropouts are identities)
    def __init__(self):
        super().__init__()
        self.backbone = resnet18_trunk()
        self.pattern_weight_layer = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Flatten(), nn.ReLU(),
                                                  nn.Linear(512, 8, bias=False))
        self.pattern_module = _Patterns()
        self.att_feature_layer = nn.Sequential(nn.Linear(1024, 512, bias=False), nn.ReLU(), nn.Dropout(0.5))
        self.att_pred_layer = nn.Sequential(nn.Linear(512, 6, bias=False), nn.Sigmoid())
        self.com_feature_layer = nn.Sequential(nn.Linear(1024, 512, bias=False), nn.ReLU(), nn.Dropout(0.5))  # unused in forward
        self.alpha_predict_layer = nn.Sequential(nn.Linear(1024, 2, bias=False), nn.Sigmoid())               # unused in forward
        self.com_pred_layer = nn.Sequential(nn.Linear(1024, 1024, bias=False), nn.ReLU(), nn.Dropout(0.5),
                                            nn.Linear(1024, 512, bias=False), nn.ReLU(),
                                            nn.Linear(512, 5, bias=False), nn.Softmax(dim=1))

    def forward(self, x, saliency):
        fm = self.backbone(x)
        pw = self.pattern_weight_layer(fm)
        sal_down = F.max_pool2d(F.max_pool2d(saliency, 3, 2, 1), 3, 2, 1)
        feat = self.pattern_module(fm, sal_down, pw)
        attrs = self.att_pred_layer(self.att_feature_layer(feat))
        dist = self.com_pred_layer(feat)
        return pw, attrs, dist


def samp_postprocess(pw, attrs, dist):
    """SAMPNetScorer.score post-processing, samp_net.py:957-989, for one image (1-D tensors/arrays)."""
    pw_np = F.softmax(torch.as_tensor(pw), dim=0).numpy()
    sd = np.asarray(dist, dtype=np.float32)
    idx = int(np.argmax(pw_np))
    raw = float(np.sum(np.array([1, 2, 3, 4, 5]) * sd))
    comp = max(0.0, min(10.0, (raw - 1) / 4.0 * 10.0))
    return {
        'comp_score': round(comp, 2), 'raw_score': round(raw, 2), 'pattern': COMPOSITION_PATTERNS[idx],
        'pattern_index': idx,
        'pattern_weights': {COMPOSITION_PATTERNS[i]: float(pw_np[i]) for i in range(8)},
        'score_distribution': sd.tolist(), 'attributes': np.asarray(attrs, dtype=np.float32).tolist(),
        'power_point_score': round(comp / 2, 2),
    }
