"""Dropout randomness of the HIP path.

The reference draws dropout masks from torch's global generator, re-seeded every update with
`seed + num_updates` (trainer.py:934-938 -> utils.set_torch_seed).  Here every dropout SITE (one call of
FairseqDropout / one fused epilogue / one attention-probability dropout) gets a 32-bit key derived from
(that same seed, the site's ordinal inside the step), and the kernels evaluate keep(key, element index)
on the fly (csrc/cst_common.h: cst_drop_bits) — forward and backward regenerate the same mask, nothing is
stored.  `keep_mask_numpy` is the same function in numpy (used by the tests as the checker)."""
import numpy as np

_M32 = 0xFFFFFFFF


def _hash32(x):
    x &= _M32
    x ^= x >> 16
    x = (x * 0x7FEB352D) & _M32
    x ^= x >> 15
    x = (x * 0x846CA68B) & _M32
    x ^= x >> 16
    return x


class DropoutState:
    """Per-process dropout stream: reseed(seed) at the start of every update, next_key() once per dropout site."""

    def __init__(self):
        self.seed = 1
        self.site = 0

    def reseed(self, seed):
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.site = 0

    def next_key(self):
        self.site += 1
        return _hash32((self.seed & _M32) ^ _hash32(self.site * 0x9E3779B1 + (self.seed >> 32)))


STATE = DropoutState()


def reseed(seed):
    STATE.reseed(seed)


def next_key():
    return STATE.next_key()


def keep_mask_numpy(key, n, p):
    """Boolean keep mask of elements [0, n) of the site `key` at drop probability p (bit-exact with cst_drop_bits)."""
    u = np.uint64
    idx = np.arange(n, dtype=np.uint64)
    pair = idx >> u(1)
    key2 = (int(key) * 0x2C1B3C6D + 0x297A2D39) & _M32
    x = (((pair & u(_M32)) ^ u(key)) + (pair >> u(32)) * u(0x9E3779B1)) & u(_M32)
    x = (x * u(0x9E3779B1)) & u(_M32)
    x = ((x ^ (x >> u(15))) + u(key2)) & u(_M32)
    x = (x * u(0x85EBCA77)) & u(_M32)
    x ^= x >> u(13)
    half = np.where((idx & u(1)) == 1, x >> u(16), x & u(0xFFFF))
    return half >= u(int(p * 65536.0 + 0.5))


def _bits32(key, key2, idx):
    """cst_drop_bits32 (csrc/cst_common.h) on uint64 arrays holding 32-bit values."""
    u = np.uint64
    x = ((idx ^ u(key)) * u(0x9E3779B1)) & u(_M32)
    x = ((x ^ (x >> u(15))) + u(key2)) & u(_M32)
    x = (x * u(0x85EBCA77)) & u(_M32)
    return x ^ (x >> u(13))


def _signatures(key, key2, n):
    """[n, 32] int8 signatures: bytes (little endian) of the 8 words cst_drop_bits32(key, key2, 8 i + w)."""
    idx = np.arange(n, dtype=np.uint64)[:, None] * np.uint64(8) + np.arange(8, dtype=np.uint64)[None, :]
    return _bits32(key, key2, idx).astype(np.uint32).view(np.int8).reshape(n, 32)


def keep_mask_attn_numpy(key, rows, Tk, p):
    """Keep mask [rows, Tk] of an attention-probability dropout site (csrc/cst_common.h, cst_asig_*: the mask every attention kernel
    takes from an i8 MFMA): row rho = (b*H + h)*Tq + q of the probability tensor and key k carry 32-byte int8 signatures,
      D = sigA(k) . sigB(rho)  (int32),   keep = int16((D << 2) & 0xffff) >= round(p * 65536) - 32768   (the low 14 bits of D)."""
    key = int(key) & _M32
    key2 = (key * 0x2C1B3C6D + 0x297A2D39) & _M32
    a = _signatures(key ^ 0x68E31DA4, key2, Tk).astype(np.int32)
    b = _signatures(key, key2, rows).astype(np.int32)
    d = b @ a.T
    h = ((d.astype(np.int64) << 2) & 0xFFFF)
    h = np.where(h >= 32768, h - 65536, h)
    return h >= int(p * 65536.0 + 0.5) - 32768
