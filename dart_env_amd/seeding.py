"""Seed plumbing with the reference's exact semantics (restated from the behaviour of reference
gym/utils/seeding.py:11-91): seed -> SHA-512 of its decimal string -> first 8 bytes as little-endian uint32 words
-> ``numpy.random.RandomState`` (MT19937, init_by_array).  Identical seeds therefore give the identical reset-noise
stream the reference's ``DartEnv.seed`` (dart_env.py:117-119) would give."""
import hashlib
import os
import struct

import numpy as np


class SeedError(ValueError):
    pass


def _words_to_int(data: bytes) -> int:
    """little-endian uint32 words -> big integer; the reference pads with 1..4 NUL bytes first."""
    data = data + b"\0" * (4 - len(data) % 4)
    words = struct.unpack("<%dI" % (len(data) // 4), data)
    return sum(w << (32 * i) for i, w in enumerate(words))


def create_seed(a=None, max_bytes=8) -> int:
    if a is None:
        return _words_to_int(os.urandom(max_bytes))
    if isinstance(a, str):
        raw = a.encode("utf8")
        raw += hashlib.sha512(raw).digest()
        return _words_to_int(raw[:max_bytes])
    if isinstance(a, (int, np.integer)):
        return int(a) % 2 ** (8 * max_bytes)
    raise SeedError("Invalid type for seed: %s (%r)" % (type(a), a))


def hash_seed(seed=None, max_bytes=8) -> int:
    if seed is None:
        seed = create_seed(max_bytes=max_bytes)
    digest = hashlib.sha512(str(seed).encode("utf8")).digest()
    return _words_to_int(digest[:max_bytes])


def int_list_from_bigint(value: int):
    if value < 0:
        raise SeedError("Seed must be non-negative, not %r" % value)
    if value == 0:
        return [0]
    out = []
    while value > 0:
        value, low = divmod(value, 2 ** 32)
        out.append(low)
    return out


def np_random(seed=None):
    """-> (RandomState, seed_used); raises for negative / non-integer seeds like the reference."""
    if seed is not None and not (isinstance(seed, (int, np.integer)) and 0 <= seed):
        raise SeedError("Seed must be a non-negative integer or omitted, not %r" % (seed,))
    seed = create_seed(seed)
    rng = np.random.RandomState()
    rng.seed(int_list_from_bigint(hash_seed(seed)))
    return rng, seed


def mt_keys(seeds):
    """The init_by_array keys ``np_random(seed_i)`` would feed MT19937, for many non-negative int seeds at once:
    -> (keys (n, 2) uint32, key_len (n,) int32).  Same words as ``int_list_from_bigint(hash_seed(create_seed(s)))``
    (the 8 digest bytes are two little-endian words; a zero high word is dropped, an all-zero value gives [0])."""
    n = len(seeds)
    raw = bytearray(8 * n)
    sha = hashlib.sha512
    for i, sd in enumerate(seeds):
        raw[8 * i:8 * i + 8] = sha(str(int(sd) % 2 ** 64).encode("utf8")).digest()[:8]
    keys = np.frombuffer(bytes(raw), dtype="<u4").reshape(n, 2).astype(np.uint32)
    key_len = np.where(keys[:, 1] != 0, 2, 1).astype(np.int32)
    return keys, key_len
