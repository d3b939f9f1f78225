"""Bit-reproducible synthetic inputs for tests and bench.py.

LCG  x <- x*1664525 + 1013904223 (mod 2^32), byte = (x >> 8) & 255.  Everything here is integer
arithmetic on numpy arrays so the same frames can be regenerated anywhere (no files, no RNG state).
"""
import numpy as np

_A = np.uint32(1664525)
_C = np.uint32(1013904223)
_jump_cache = {}


def _jump_tables(n):
    """A_k, C_k with x_k = A_k * x_0 + C_k (mod 2^32), k = 1..n."""
    have = _jump_cache.get("n", 0)
    if have < n:
        with np.errstate(over="ignore"):
            a = np.full(n, _A, dtype=np.uint32)
            A = np.multiply.accumulate(a, dtype=np.uint32)                      # a^k, k=1..n
            geo = np.concatenate([np.ones(1, np.uint32), A[:-1]])               # a^0..a^(n-1)
            Cs = np.add.accumulate(geo, dtype=np.uint32) * _C                   # c*(1+a+..+a^(k-1))
        _jump_cache.update(n=n, A=A, C=Cs.astype(np.uint32))
    return _jump_cache["A"][:n], _jump_cache["C"][:n]


def lcg_states(seed, n):
    """The n LCG states following `seed` (uint32 array)."""
    A, Cc = _jump_tables(n)
    with np.errstate(over="ignore"):
        return A * np.uint32(seed & 0xFFFFFFFF) + Cc


def lcg_bytes(seed, n):
    return ((lcg_states(seed, n) >> np.uint32(8)) & np.uint32(255)).astype(np.uint8)


def corners_frame(seed, w=640, h=480, block=8):
    """'corners' frame: block x block LCG-valued tiles -> 3x3 integer box blur (round half up, edge
    replicate) -> + ((x>>8) % 9) - 4 per-pixel LCG noise -> clamp.  Yields >> 1000 FAST-20 corners."""
    bw, bh = (w + block - 1) // block, (h + block - 1) // block
    st = lcg_states(seed, bw * bh + w * h)
    tiles = ((st[:bw * bh] >> np.uint32(8)) & np.uint32(255)).astype(np.int32).reshape(bh, bw)
    img = np.repeat(np.repeat(tiles, block, axis=0), block, axis=1)[:h, :w]
    p = np.pad(img, 1, mode="edge")
    s = sum(p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3))
    img = (s + 4) // 9
    noise = ((st[bw * bh:] >> np.uint32(8)) % np.uint32(9)).astype(np.int32).reshape(h, w) - 4
    return np.clip(img + noise, 0, 255).astype(np.uint8)


def noise_frame(seed, w=640, h=480):
    """pure LCG noise: FAST fires almost everywhere -> exercises retainBest ties and capacities."""
    return lcg_bytes(seed, w * h).reshape(h, w)


def constant_frame(value=128, w=640, h=480):
    return np.full((h, w), value, np.uint8)


def ramp_frame(w=640, h=480):
    return np.tile((np.arange(w) * 255 // max(w - 1, 1)).astype(np.uint8), (h, 1))


def corners_batch(first_seed, nframes, w=640, h=480):
    """frames with seeds first_seed .. first_seed+nframes-1 (seed = 1 + frame index by convention)."""
    return np.stack([corners_frame(first_seed + i, w, h) for i in range(nframes)])


def random_descriptors(seed, n, nbytes=32):
    return lcg_bytes(seed, n * nbytes).reshape(n, nbytes)


def perturbed_descriptors(desc, seed, flip_prob_256=26, replace_frac_256=77, return_mask=False):
    """keyframe k+1 from keyframe k: each bit flipped w.p. flip_prob_256/256 (~0.1), and rows replaced
    w.p. replace_frac_256/256 (~0.3) (config #4 generator)."""
    n, nb = desc.shape
    st = lcg_bytes(seed, n * nb * 8 + n + n * nb)
    flips = (st[:n * nb * 8] < flip_prob_256).reshape(n, nb, 8)
    mask = np.zeros((n, nb), np.uint8)
    for b in range(8):
        mask |= (flips[:, :, b].astype(np.uint8) << b)
    out = desc ^ mask
    repl = st[n * nb * 8:n * nb * 8 + n] < replace_frac_256
    fresh = st[n * nb * 8 + n:].reshape(n, nb)
    out[repl] = fresh[repl]
    return (out, repl) if return_mask else out


def keyframe_table(nkeyframes, cap=1000, seed=7):
    """config #4 table: K keyframes x cap x 32-byte descriptors, keyframe k+1 = keyframe k with each bit
    flipped w.p. ~0.1 and ~30 % of the rows replaced.  Angles (degrees) follow the keyframes: a common rotation of
    -10..10 degrees per step plus -1..1 degree of per-feature jitter; replaced rows get a fresh angle.
    Returns (table uint8 [K, cap, 32], angles float32 [K, cap], counts int32 [K])."""
    table = np.zeros((nkeyframes, cap, 32), np.uint8)
    angles = np.zeros((nkeyframes, cap), np.float32)
    d = random_descriptors(seed, cap)
    a = (lcg_states(seed + 1, cap) >> np.uint32(8)) % np.uint32(36000)
    a = a.astype(np.int64)                                    # centi-degrees: exact integer bookkeeping
    for k in range(nkeyframes):
        table[k] = d
        angles[k] = (a.astype(np.float32) / np.float32(100.0))
        nd, replaced = perturbed_descriptors(d, 100003 + 17 * k + seed, return_mask=True)
        st = lcg_states(200003 + 31 * k + seed, 2 * cap + 1)
        rot = int((st[0] >> np.uint32(8)) % np.uint32(2001)) - 1000
        jit = ((st[1:cap + 1] >> np.uint32(8)) % np.uint32(201)).astype(np.int64) - 100
        fresh = ((st[cap + 1:] >> np.uint32(8)) % np.uint32(36000)).astype(np.int64)
        a = (a + rot + jit) % 36000
        a = np.where(replaced, fresh, a)
        d = nd
    return table, angles, np.full(nkeyframes, cap, np.int32)
