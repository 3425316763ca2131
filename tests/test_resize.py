"""scipy.misc.imresize (dataset_.py:481-495; = PIL bilinear on uint8) restated in the oracle, pinned bit for bit against the
installed Pillow -- the third-party library the reference called, NOT the reference -- on random and structured images, up- and
down-scaling, odd sizes, and the reference's 240x320 -> 227x227 case."""
import numpy as np
import pytest

from oracle import lrcn_oracle as O

PIL = pytest.importorskip("PIL.Image")


@pytest.mark.parametrize("src,dst", [((240, 320), (227, 227)), ((240, 320), (240, 320)), ((37, 53), (80, 90)), ((80, 90), (37, 53)),
                                     ((64, 64), (64, 31)), ((31, 64), (63, 64)), ((10, 10), (1, 1)), ((3, 5), (240, 320)),
                                     ((480, 640), (240, 320)), ((100, 7), (13, 200))])
def test_imresize_restatement_equals_pillow(src, dst):
    rng = np.random.default_rng(src[0] * 1000 + dst[1])
    for kind in range(3):
        if kind == 0:
            img = rng.integers(0, 256, src + (3,), dtype=np.uint8)
        elif kind == 1:
            yy, xx = np.mgrid[0:src[0], 0:src[1]]
            img = np.stack([(yy * 7 + xx * 3) % 256, (xx * xx // 5) % 256, 255 - (yy + xx) % 256], axis=2).astype(np.uint8)
        else:
            img = (rng.integers(0, 2, src + (3,)) * 255).astype(np.uint8)           # extremes: exercises the clip
        want = np.asarray(PIL.fromarray(img).resize((dst[1], dst[0]), resample=PIL.BILINEAR))
        got = O.imresize_bilinear_u8(img, dst)
        assert got.dtype == np.uint8 and got.shape == dst + (3,)
        assert np.array_equal(got, want), "kind %d: %d pixels differ, max %d" % (kind, int((got != want).sum()),
                                                                              int(np.abs(got.astype(int) - want.astype(int)).max()))


def test_coefficients_are_normalised():
    for a, b in ((320, 227), (240, 227), (53, 90), (640, 320)):
        bounds, kk, ksize = O.pil_bilinear_coeffs(a, b)
        assert kk.shape == (b, ksize) and (bounds[:, 0] >= 0).all() and (bounds[:, 0] + bounds[:, 1] <= a).all()
        assert np.abs(kk.sum(axis=1) - (1 << 22)).max() <= ksize                          # rounding of each tap, at most 1/2 unit each
