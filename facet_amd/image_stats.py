"""GPU-backed stand-ins for reference analyzers/image_cache.py `ImageCache` and analyzers/technical.py `TechnicalAnalyzer`
(the *_data / detect_* / get_* methods multi_pass.py:425-444 and batch_processor.py call with `cache=`) - SURVEY 8(f)-1.

The reference builds an ImageCache per image on the CPU (two cv2.cvtColor, one cv2.Laplacian) and every metric then re-scans
the arrays (calcHist, percentile, filter2D, mean, std). Here ONE engine call (fe_image_stats) scans a whole resident batch and
returns 264 numbers per image - gray histogram, Laplacian sum / sum of squares, |Immerkaer| sum, saturation sum, hue-saturation
entropy term - and every metric below is closed-form host arithmetic on that record:
  variance = E[x^2] - E[x]^2 on exact integer sums; percentiles / std / means from the 256-bin histogram (exactly numpy's
  linear-interpolation percentile, since order statistics of uint8 data are determined by the histogram);
  entropy = log2(N) - sum(c log2 c)/N.
Method names, arguments, dict keys and rounding follow the reference; `cache` must be an `ImageCache` made by
`ImageCache.from_batch` / `ImageCache(img, engine=...)`. cv2 semantics are restated [DEP-KNOWLEDGE], see kernels_stats.hip.
"""
import struct

import numpy as np

from ._lib import FE_STATS_DOUBLES

_H, _LAP_S, _LAP_SS, _NOISE, _SAT, _CLOGC = slice(0, 256), 256, 257, 258, 259, 260


class ImageCache:
    """Holds the statistics record of one image (and optionally its gray / hsv planes)."""
    __slots__ = ['stats', 'height', 'width', 'gray', 'hsv', 'laplacian_variance', '_cum']

    def __init__(self, img_cv=None, engine=None, _record=None, _shape=None, _gray=None, _hsv=None, keep_planes=False):
        if _record is None:
            if engine is None:          # reference signature ImageCache(img_cv): use the process-wide engine
                from . import default_engine
                engine = default_engine()
            st, g, hv = engine.image_stats(np.asarray(img_cv)[None], want_gray=keep_planes, want_hsv=keep_planes)
            _record, _shape = st[0], img_cv.shape[:2]
            _gray, _hsv = (g[0], hv[0]) if keep_planes else (None, None)
        assert _record.shape == (FE_STATS_DOUBLES,)
        self.stats = _record
        self.height, self.width = int(_shape[0]), int(_shape[1])
        self.gray, self.hsv = _gray, _hsv
        n = float(self.height * self.width)
        mean = _record[_LAP_S] / n
        self.laplacian_variance = _record[_LAP_SS] / n - mean * mean
        self._cum = None

    @classmethod
    def from_batch(cls, engine, images, keep_planes=False):
        """images: BGR uint8 [n,h,w,3] (or a device tuple) -> list[ImageCache], one engine call."""
        st, g, hv = engine.image_stats(images, want_gray=keep_planes, want_hsv=keep_planes)
        shape = images[2:4] if isinstance(images, tuple) else np.asarray(images).shape[1:3]
        return [cls(_record=st[i], _shape=shape, _gray=None if g is None else g[i], _hsv=None if hv is None else hv[i]) for i in range(st.shape[0])]

    # -- order statistics of the gray plane from its histogram ---------------------------------------------------------
    def _order_stat(self, k):
        if self._cum is None:
            self._cum = np.cumsum(self.stats[_H])
        return int(np.searchsorted(self._cum, k, side='right'))

    def percentile(self, q):
        """np.percentile(gray, q) (method='linear') for the uint8 plane."""
        n = self.height * self.width
        virt = (q / 100.0) * (n - 1)
        lo = int(np.floor(virt))
        t = virt - lo
        a, b = float(self._order_stat(lo)), float(self._order_stat(min(lo + 1, n - 1)))
        d = b - a
        return b - d * (1 - t) if t >= 0.5 else a + d * t


class TechnicalAnalyzer:
    @staticmethod
    def _need(cache, image_cv=None):
        """The reference recomputes gray / hsv with cv2 when cache is None; here that means one GPU scan of the image."""
        if cache is None and image_cv is not None and not isinstance(image_cv, bool):
            return ImageCache(image_cv)
        if not isinstance(cache, ImageCache):
            raise TypeError("pass cache=ImageCache(...) built by facet_amd.image_stats (the GPU computes the statistics)")
        return cache

    # -- the three cache-less conveniences of the reference (technical.py:29-36, 60-78, 118-127) ---------------------
    @staticmethod
    def get_sharpness(image_cv):
        return 0 if image_cv is None else min(10.0, ImageCache(image_cv).laplacian_variance / 50.0)

    @staticmethod
    def get_color_harmony(image_cv):
        return TechnicalAnalyzer.get_color_harmony_data(image_cv)['normalized']

    @staticmethod
    def get_exposure_score(image_cv):
        c = ImageCache(image_cv)
        n = float(c.height * c.width)
        clipped = c.stats[_H][:6].sum() / n + c.stats[_H][250:].sum() / n       # gray <= 5, gray >= 250
        return max(0, 10 - clipped * 10)

    @staticmethod
    def get_sharpness_data(image_cv, cache=None):
        if image_cv is None:
            return {'raw_variance': 0, 'normalized': 0}
        v = TechnicalAnalyzer._need(cache, image_cv).laplacian_variance
        return {'raw_variance': v, 'normalized': float(min(10.0, v / 50.0))}

    @staticmethod
    def get_color_harmony_data(image_cv, cache=None):
        if image_cv is None:
            return {'raw_entropy': 0, 'normalized': 0}
        c = TechnicalAnalyzer._need(cache, image_cv)
        n = float(c.height * c.width)
        ent = np.log2(n) - c.stats[_CLOGC] / n if n > 0 else 0
        return {'raw_entropy': ent, 'normalized': float(min(10.0, ent * 10.0 / 15.5))}

    @staticmethod
    def get_histogram_data(image_cv, shadow_threshold=0.15, highlight_threshold=0.10, cache=None):
        if image_cv is None:
            return {'histogram_bytes': None, 'spread': 0, 'mean_luminance': 0.5, 'bimodality': 0, 'exposure_score': 5.0,
                    'shadow_clipped': 0, 'highlight_clipped': 0, 'is_silhouette': 0}
        c = TechnicalAnalyzer._need(cache, image_cv)
        counts = c.stats[_H].astype(np.float32)            # cv2.calcHist returns float32 counts
        total = counts.sum()
        dist = counts / total if total > 0 else counts
        levels = np.arange(256)
        centre = np.sum(levels * dist)
        spread = np.sqrt(np.sum(((levels - centre) ** 2) * dist))
        lum = centre / 255.0
        dark, bright = np.sum(dist[:30]), np.sum(dist[225:])
        silhouette = 1 if (np.sum(dist[:85]) > 0.35 and np.sum(dist[170:]) > 0.25) else 0
        try:
            from scipy.stats import kurtosis
            bimodality = -kurtosis(dist * 256, fisher=True)
        except (ImportError, ValueError):
            bimodality = 0
        clip_cost = 0 if silhouette else dark * 4.0 + bright * 5.0
        score = 7.0 - abs(lum - 0.5) * 8 + min(4.0, spread / 20.0) - max(0, bimodality - 1.0) * 0.6 - clip_cost
        return {'histogram_bytes': struct.pack('256f', *dist), 'spread': round(spread, 4), 'mean_luminance': round(lum, 4),
                'bimodality': round(bimodality, 4), 'exposure_score': round(max(0, min(10.0, score)), 2),
                'shadow_clipped': 1 if dark > shadow_threshold else 0, 'highlight_clipped': 1 if bright > highlight_threshold else 0,
                'is_silhouette': silhouette}

    @staticmethod
    def detect_monochrome(image_cv, threshold=0.1, cache=None):
        if image_cv is None:
            return {'is_monochrome': 0, 'mean_saturation': 0}
        c = TechnicalAnalyzer._need(cache, image_cv)
        sat = (c.stats[_SAT] / (c.height * c.width)) / 255.0
        return {'is_monochrome': 1 if sat < threshold else 0, 'mean_saturation': round(sat, 4)}

    @staticmethod
    def get_dynamic_range(image_cv, cache=None):
        if image_cv is None:
            return {'dynamic_range_stops': 0}
        c = TechnicalAnalyzer._need(cache, image_cv)
        lo, hi = c.percentile(2), c.percentile(98)
        lo = 1 if lo < 1 else lo
        return {'dynamic_range_stops': round(np.log2(max(hi, 1) / lo), 2)}

    @staticmethod
    def get_noise_estimate(image_cv, cache=None):
        if image_cv is None:
            return {'noise_sigma': 0}
        c = TechnicalAnalyzer._need(cache, image_cv)
        sigma = c.stats[_NOISE] * np.sqrt(0.5 * np.pi) / (6 * (c.width - 2) * (c.height - 2))
        return {'noise_sigma': round(sigma, 2)}

    @staticmethod
    def get_contrast_score(image_cv, cache=None):
        if image_cv is None:
            return {'contrast_score': 0, 'percentile_contrast': 0, 'rms_contrast': 0}
        c = TechnicalAnalyzer._need(cache, image_cv)
        span = (c.percentile(95) - c.percentile(5)) / 255.0
        counts, levels = c.stats[_H], np.arange(256, dtype=np.float64)
        n = counts.sum()
        mu = (counts * levels).sum() / n
        rms = np.sqrt((counts * (levels - mu) ** 2).sum() / n) / 255.0
        return {'contrast_score': round(min(10.0, span * 5.0 + rms * 20.0), 2), 'percentile_contrast': round(span, 4), 'rms_contrast': round(rms, 4)}

    @staticmethod
    def analyze_batch(engine, images, shadow_threshold=0.15, highlight_threshold=0.10, mono_threshold=0.1):
        """The seven dicts multi_pass.py:425-444 computes per image, for a whole BGR batch with one engine call."""
        T = TechnicalAnalyzer
        out = []
        for c in ImageCache.from_batch(engine, images):
            out.append({'sharpness': T.get_sharpness_data(True, c), 'color': T.get_color_harmony_data(True, c),
                        'histogram': T.get_histogram_data(True, shadow_threshold, highlight_threshold, c),
                        'mono': T.detect_monochrome(True, mono_threshold, c), 'dynamic_range': T.get_dynamic_range(True, c),
                        'noise': T.get_noise_estimate(True, c), 'contrast': T.get_contrast_score(True, c), 'cache': c})
        return out
