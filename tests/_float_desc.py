"""float descriptor rows derived from binary ones (test data for the float paths of the matchers)"""
import numpy as np


def floaten(d32, dim, real):
    """float rows with the neighbourhood structure of the 32-byte ones: the first `dim` bits as 0.0 / 1.0 (real = False: L2^2 = the Hamming
    distance over those bits - equal distances everywhere, which is what exercises the visiting-order tie-break of the keys) or the bits
    scaled, plus a deterministic fraction per element (real = True: distances with full float mantissas, where the summation order shows)"""
    bits = np.unpackbits(np.ascontiguousarray(d32, np.uint8), axis=1)[:, :dim].astype(np.float32)
    if not real:
        return np.ascontiguousarray(bits)
    n = len(bits)
    frac = ((np.arange(n * dim, dtype=np.uint64).reshape(n, dim) * np.uint64(2654435761) + np.uint64(12345)) % np.uint64(1 << 20)).astype(np.float32)
    rows = np.ascontiguousarray(d32, np.uint8).astype(np.uint64).sum(1, keepdims=True)  # a per-row phase so that equal rows of different sets differ
    frac = (frac + (rows * np.uint64(977) % np.uint64(1 << 20)).astype(np.float32)) % np.float32(1 << 20)
    return np.ascontiguousarray(bits * np.float32(0.75) + frac * np.float32(0.2 / (1 << 20)), np.float32)
