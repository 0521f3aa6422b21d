"""Oracle: the synthetic unlabeled pool of BASELINE configs[3] ("pool of 10 000 synthetic 512^2 images generated
on-device from Philox(seed=20, image_id)").

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  numpy restatement of aod_meh_hua_amd/csrc/elementwise.hip synth_normal_kernel:
element block i (four consecutive fp32 values of image `image_id`, flattened [3, H, W]) comes from ONE Philox4x32-10 block at counter
(i_lo, i_hi, id_lo, id_hi) under key (seed_lo, seed_hi): two Box-Muller pairs (r_a cos 2 pi u_1, r_a sin 2 pi u_1, r_b cos 2 pi u_3,
r_b sin 2 pi u_3) with r = sqrt(-2 ln u).  There is nothing of the reference to follow here (its pool is VOC trainval,
tools/train_RetinaNet.py:221-225); the oracle pins that a pool image is a pure function of (seed, image id)."""
import numpy as np

from .hua import _u01, philox4x32


def philox_normal_image(seed, image_id, n_elems):
    """-> float32 [n_elems] (n_elems % 4 == 0), computed in float64 and rounded: the HIP kernel uses the hardware's 1-ulp
    transcendentals in fp32, so it agrees to a few fp32 ulps of the result's scale, not bit for bit."""
    assert n_elems % 4 == 0
    i = np.arange(n_elems // 4, dtype=np.uint64)
    r = philox4x32((i & np.uint64(0xffffffff)).astype(np.uint32), (i >> np.uint64(32)).astype(np.uint32),
                   np.uint32(image_id & 0xffffffff), np.uint32((image_id >> 32) & 0xffffffff), seed & 0xffffffff, (seed >> 32) & 0xffffffff)
    u = [_u01(x).astype(np.float64) for x in r]
    ra, rb = np.sqrt(-2.0 * np.log(u[0])), np.sqrt(-2.0 * np.log(u[2]))
    out = np.stack([ra * np.cos(2 * np.pi * u[1]), ra * np.sin(2 * np.pi * u[1]), rb * np.cos(2 * np.pi * u[3]), rb * np.sin(2 * np.pi * u[3])], 1)
    return out.reshape(-1).astype(np.float32)
