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


def keep_mask_attn_numpy(key, rows, Tk, p):
    """Keep mask [rows, Tk] of an attention-probability dropout site (csrc/cst_common.h: cst_adrop_* — the SEPARABLE mask all
    attention kernels evaluate): row rho = (b*H + h)*Tq + q of the probability tensor, key k.
      R1, R2 = two hashes of rho;  C = a hash of the key pair k >> 1;  x = (R1 ^ C) & 0xffffff;  t = x * 0x9E3779;  t ^= t >> 15;
      w = (t & 0xffffff) * 0x85EBCB + R2;  keep = int16(half (k & 1 ? high : low) of w) >= round(p * 65536) - 32768."""
    u = np.uint64
    key = int(key) & _M32
    key2 = (key * 0x2C1B3C6D + 0x297A2D39) & _M32
    rho = np.arange(rows, dtype=np.uint64)
    r1 = _bits32(key, key2, rho)
    r2 = _bits32(key2 ^ 0xA511E9B3, key, rho)
    kp = np.arange((Tk + 1) // 2, dtype=np.uint64)
    c = _bits32(key ^ 0x68E31DA4, key2, kp)
    x = (r1[:, None] ^ c[None, :]) & u(0xFFFFFF)
    t = (x * u(0x9E3779)) & u(_M32)
    t ^= t >> u(15)
    w = ((t & u(0xFFFFFF)) * u(0x85EBCB) + r2[:, None]) & u(_M32)
    thr = int(p * 65536.0 + 0.5)
    lo = (w & u(0xFFFF)) ^ u(0x8000)   # signed compare of the half == unsigned compare with the sign bit flipped
    hi = (w >> u(16)) ^ u(0x8000)
    keep = np.empty((rows, 2 * len(kp)), dtype=bool)
    keep[:, 0::2] = lo >= u(thr)
    keep[:, 1::2] = hi >= u(thr)
    return keep[:, :Tk]
