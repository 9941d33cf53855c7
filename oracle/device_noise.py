"""TEST INFRASTRUCTURE (only tests/ may import this): CPU restatement of the Exp(1) noise the HIP sampler draws on the device when the
caller supplies none (`noise="device"`; chatttsplus_amd/csrc/sampler.hip `device_exp_noise`).  The reference draws its noise inside
torch.multinomial (gpt.py:480-481: argmax(p / q), q ~ Exp(1)) from torch's generator; the device stream replaces THAT generator for
batches the host generator cannot feed -- any i.i.d. Exp(1) stream samples the same distribution, so what has to be pinned is that the device
stream IS i.i.d. Exp(1) and keyed as documented:

    (x0, x1, x2, x3) = Philox4x32-10(counter = (element | stream << 24, uid_lo, uid_hi, step | attempt << 20), key = (seed_lo, seed_hi))
    u = ((x0 >> 8) + 0.5) / 2^24          q = -log(u)        (fp32)

Philox4x32-10: Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11), the Random123 constants."""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(counter, key):
    """counter: 4 arrays (or ints) of uint32 values, key: 2 ints.  Returns 4 uint32 arrays."""
    c = [np.asarray(x, dtype=np.uint64) & MASK for x in counter]
    c = list(np.broadcast_arrays(*c))
    k0, k1 = int(key[0]) & MASK, int(key[1]) & MASK
    for _ in range(10):
        p0 = np.uint64(M0) * c[0]
        p1 = np.uint64(M1) * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & np.uint64(MASK)
        hi1, lo1 = p1 >> np.uint64(32), p1 & np.uint64(MASK)
        c = [hi1 ^ c[1] ^ np.uint64(k0), lo1, hi0 ^ c[3] ^ np.uint64(k1), lo0]
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return [x.astype(np.uint32) for x in c]


def exp_noise(seed: int, utt_id: int, stream: int, step: int, attempt: int, n: int) -> np.ndarray:
    """q[0..n) of one multinomial row: stream = codebook 0..3 (code mode) or 4 (refine-text row)."""
    j = np.arange(n, dtype=np.uint64)
    x0 = philox4x32_10((j | np.uint64(stream << 24), utt_id & MASK, (utt_id >> 32) & MASK, (step | (attempt << 20)) & MASK),
                       (seed & MASK, (seed >> 32) & MASK))[0]
    u = ((x0 >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    return (-np.log(u.astype(np.float32))).astype(np.float32)
