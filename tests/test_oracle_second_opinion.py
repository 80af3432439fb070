"""Second opinion for the 'parity unpinned' CLIP oracle (oracle/clip_vit.py): the same weights through HuggingFace transformers'
independent CLIP implementation (CLIPVisionModelWithProjection / CLIPTextModelWithProjection, installed offline, randomly
initialised from a config - no download). open_clip itself, the reference's dependency (requirements.txt:8), is absent; HF CLIP is
the other widely used implementation of the published architecture, with its own parameter naming and attention code, so
agreement shows the restatement is the standard ViT / text transformer and not a private variant. CPU only, reduced widths plus
one pass at the real ViT-L/14 geometry (24 x 1024, 257 tokens).
"""
import numpy as np
import pytest
import torch

from oracle.clip_vit import TextTransformer, VisionTransformer

transformers = pytest.importorskip("transformers")


def _fill(module, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in module.named_parameters():
            if p.dim() >= 2:
                p.copy_(torch.randn(p.shape, generator=g) * (0.5 / np.sqrt(p.shape[-1])))
            elif "ln" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.1 * torch.randn(p.shape, generator=g))
            else:
                p.copy_(0.05 * torch.randn(p.shape, generator=g))


def _copy_blocks(src_blocks, dst_layers, d):
    with torch.no_grad():
        for b, l in zip(src_blocks, dst_layers):
            l.layer_norm1.weight.copy_(b.ln_1.weight); l.layer_norm1.bias.copy_(b.ln_1.bias)
            l.layer_norm2.weight.copy_(b.ln_2.weight); l.layer_norm2.bias.copy_(b.ln_2.bias)
            w, bias = b.attn.in_proj_weight, b.attn.in_proj_bias
            for k, proj in enumerate((l.self_attn.q_proj, l.self_attn.k_proj, l.self_attn.v_proj)):
                proj.weight.copy_(w[k * d:(k + 1) * d]); proj.bias.copy_(bias[k * d:(k + 1) * d])
            l.self_attn.out_proj.weight.copy_(b.attn.out_proj.weight); l.self_attn.out_proj.bias.copy_(b.attn.out_proj.bias)
            l.mlp.fc1.weight.copy_(b.mlp.c_fc.weight); l.mlp.fc1.bias.copy_(b.mlp.c_fc.bias)
            l.mlp.fc2.weight.copy_(b.mlp.c_proj.weight); l.mlp.fc2.bias.copy_(b.mlp.c_proj.bias)


@pytest.mark.parametrize("width,layers,heads,patch,grid,out_dim,n", [(64, 2, 4, 8, 4, 32, 3), (128, 3, 2, 14, 16, 48, 2), (1024, 24, 16, 14, 16, 768, 1)])
def test_vision_tower_equals_hf_clip(width, layers, heads, patch, grid, out_dim, n):
    ours = VisionTransformer(width, layers, heads, patch, grid, out_dim).eval()
    _fill(ours, 7)
    cfg = transformers.CLIPVisionConfig(hidden_size=width, intermediate_size=4 * width, num_hidden_layers=layers, num_attention_heads=heads,
                                        image_size=patch * grid, patch_size=patch, projection_dim=out_dim, hidden_act="gelu", layer_norm_eps=1e-5,
                                        attn_implementation="eager")
    hf = transformers.CLIPVisionModelWithProjection(cfg).eval()
    vm = hf.vision_model
    with torch.no_grad():
        vm.embeddings.patch_embedding.weight.copy_(ours.conv1.weight)
        vm.embeddings.class_embedding.copy_(ours.class_embedding)
        vm.embeddings.position_embedding.weight.copy_(ours.positional_embedding)
        vm.pre_layrnorm.weight.copy_(ours.ln_pre.weight); vm.pre_layrnorm.bias.copy_(ours.ln_pre.bias)
        vm.post_layernorm.weight.copy_(ours.ln_post.weight); vm.post_layernorm.bias.copy_(ours.ln_post.bias)
        hf.visual_projection.weight.copy_(ours.proj.t())
    _copy_blocks(ours.transformer.resblocks, vm.encoder.layers, width)
    x = torch.randn(n, 3, patch * grid, patch * grid, generator=torch.Generator().manual_seed(1))
    with torch.no_grad():
        a = ours(x)
        b = hf(pixel_values=x).image_embeds
    assert a.shape == b.shape == (n, out_dim)
    assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max())


def test_text_tower_equals_hf_clip():
    width, layers, heads, ctx, vocab, out_dim = 64, 2, 4, 16, 100, 32
    ours = TextTransformer(width, layers, heads, ctx, vocab, out_dim).eval()
    _fill(ours, 9)
    cfg = transformers.CLIPTextConfig(vocab_size=vocab, hidden_size=width, intermediate_size=4 * width, num_hidden_layers=layers,
                                      num_attention_heads=heads, max_position_embeddings=ctx, projection_dim=out_dim, hidden_act="gelu",
                                      layer_norm_eps=1e-5, eos_token_id=2, attn_implementation="eager")   # eos id 2: pool at argmax(ids), like open_clip
    hf = transformers.CLIPTextModelWithProjection(cfg).eval()
    tm = hf.text_model
    with torch.no_grad():
        tm.embeddings.token_embedding.weight.copy_(ours.token_embedding.weight)
        tm.embeddings.position_embedding.weight.copy_(ours.positional_embedding)
        tm.final_layer_norm.weight.copy_(ours.ln_final.weight); tm.final_layer_norm.bias.copy_(ours.ln_final.bias)
        hf.text_projection.weight.copy_(ours.text_projection.t())
    _copy_blocks(ours.transformer.resblocks, tm.encoder.layers, width)
    rng = np.random.default_rng(3)
    ids = rng.integers(3, vocab - 1, (4, ctx))
    for r, eot in enumerate((5, 15, 9, 1)):            # EOT = the largest id, at different positions
        ids[r, eot] = vocab - 1
        ids[r, eot + 1:] = 0
    ids = torch.from_numpy(ids)
    with torch.no_grad():
        a = ours(ids)
        b = hf(input_ids=ids).text_embeds
    assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max())


def test_resnet50_pyramid_equals_hf_resnet():
    """oracle/resnet.py ResNet50Features (the TOPIQ backbone restatement) vs transformers' ResNetModel (v1.5: stride on the 3x3)."""
    from oracle.resnet import ResNet50Features
    ours = ResNet50Features().eval()
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for m in ours.modules():
            if isinstance(m, torch.nn.Conv2d):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * np.sqrt(2.0 / (m.weight.shape[1] * m.weight.shape[2] * m.weight.shape[3])))
            elif isinstance(m, torch.nn.BatchNorm2d):
                m.weight.copy_(0.5 + 0.5 * torch.rand(m.weight.shape, generator=g)); m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
                m.running_mean.copy_(0.1 * torch.randn(m.bias.shape, generator=g)); m.running_var.copy_(0.5 + torch.rand(m.bias.shape, generator=g))
    cfg = transformers.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3], layer_type="bottleneck",
                                    hidden_act="relu", downsample_in_first_stage=False, downsample_in_bottleneck=False)
    hf = transformers.ResNetModel(cfg).eval()

    def put(conv_layer, conv, bn):
        with torch.no_grad():
            conv_layer.convolution.weight.copy_(conv.weight)
            n = conv_layer.normalization
            n.weight.copy_(bn.weight); n.bias.copy_(bn.bias); n.running_mean.copy_(bn.running_mean); n.running_var.copy_(bn.running_var)

    put(hf.embedder.embedder, ours.conv1, ours.bn1)
    for stage, layer in zip(hf.encoder.stages, (ours.layer1, ours.layer2, ours.layer3, ours.layer4)):
        for hb, ob in zip(stage.layers, layer):
            put(hb.layer[0], ob.conv1, ob.bn1); put(hb.layer[1], ob.conv2, ob.bn2); put(hb.layer[2], ob.conv3, ob.bn3)
            if ob.downsample is not None:
                put(hb.shortcut, ob.downsample[0], ob.downsample[1])
    x = torch.randn(2, 3, 96, 128, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        feats = ours(x)
        out = hf(pixel_values=x, output_hidden_states=True)
        stem = hf.embedder.embedder(x)
    assert float((feats[0] - stem).abs().max()) <= 1e-5 * float(stem.abs().max())
    for f, hs in zip(feats[1:], out.hidden_states[1:]):
        assert f.shape == hs.shape and float((f - hs).abs().max()) <= 1e-4 * float(hs.abs().max())


def test_cfanet_encoder_block_equals_torch_transformer_encoder_layer():
    """The CFANet head has no independent implementation to lean on here (pyiqa absent, transformers has no CFANet). Its self-attention
    block, though, is the textbook pre-norm encoder layer: the oracle's EncoderLayer (oracle/topiq.py, pyiqa key names) must equal
    torch's own nn.TransformerEncoderLayer(norm_first=True, activation='gelu') on the same weights - a second opinion on the order of
    norm / attention / residual / MLP, which a checkpoint cannot reveal (these choices carry no parameters)."""
    import torch.nn as nn
    from oracle.topiq import EncoderLayer
    d, heads, ff, L, B = 64, 4, 256, 19, 3
    ours = EncoderLayer(d, heads, ff).eval()
    g = torch.Generator().manual_seed(3)
    for p in ours.parameters():
        p.data = torch.randn(p.shape, generator=g) * 0.1
    ref = nn.TransformerEncoderLayer(d, heads, dim_feedforward=ff, dropout=0.0, activation="gelu", norm_first=True).eval()
    sd = ours.state_dict()
    ref.load_state_dict({k: sd[k] for k in ref.state_dict()})          # identical key names: self_attn.*, linear1/2.*, norm1/2.*
    x = torch.randn(L, B, d, generator=g)
    with torch.no_grad():
        a = ours(x)
        b = ref(x)
    assert float((a - b).abs().max()) < 1e-5
