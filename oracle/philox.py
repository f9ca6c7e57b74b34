"""Philox4x32-10 counter-based RNG, numpy restatement (oracle; test infrastructure).

The reference draws from torch's global device generator with data-dependent
draw counts (`airgym/envs/base/hovering.py:256,316-329,350-353`), which cannot
be reproduced across back-ends.  The build replaces it with a counter-based
generator so that oracle and HIP kernel consume *identical* random numbers:

    key     = (seed_lo, seed_hi)
    counter = (global_env_id, tick, stream, block)

`tick` is the env-step index of the handle (monotonic), `stream` separates the
consumers (0 = reset draws, 1 = observation noise), `block` indexes successive
128-bit outputs.  Algorithm: Salmon et al., "Parallel Random Numbers: As Easy
as 1, 2, 3" (SC'11), Philox-4x32 with 10 rounds; known-answer vectors from the
Random123 distribution are checked in tests/test_oracle_philox.py.
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = np.uint32(0x9E3779B9)
PHILOX_W1 = np.uint32(0xBB67AE85)

STREAM_RESET = 0
STREAM_OBS_NOISE = 1
STREAM_PHASE = 2      # the build's AG_FLAG_STAGGER_PHASE opt-in: initial progress of a full reset

TWO_PI_F32 = np.float32(6.283185307179586)
INV_2_24 = np.float32(1.0 / 16777216.0)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All inputs broadcastable uint32 arrays."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint32) for c in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = PHILOX_M0 * c0.astype(np.uint64)
            p1 = PHILOX_M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(PHILOX_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(PHILOX_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def u32_to_unit_float(x):
    """[0,1) with 24 random bits: (x >> 8) * 2^-24 (exact in f32)."""
    return (np.asarray(x, dtype=np.uint32) >> np.uint32(8)).astype(np.float32) * INV_2_24


def u32_to_open_unit_float(x):
    """(0,1] : ((x >> 8) + 1) * 2^-24, used as the log argument of Box-Muller."""
    return ((np.asarray(x, dtype=np.uint32) >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * INV_2_24


def raw_blocks(seed, env_ids, tick, stream, nblocks):
    """uint32 array [N, 4*nblocks] of raw Philox outputs for the given counters."""
    env_ids = np.asarray(env_ids, dtype=np.uint32)
    k0 = np.uint32(seed & 0xFFFFFFFF)
    k1 = np.uint32((seed >> 32) & 0xFFFFFFFF)
    out = np.empty((env_ids.shape[0], 4 * nblocks), dtype=np.uint32)
    for b in range(nblocks):
        r = philox4x32_10(env_ids, np.uint32(tick & 0xFFFFFFFF), np.uint32(stream), np.uint32(b), k0, k1)
        for j in range(4):
            out[:, 4 * b + j] = r[j]
    return out


def reset_uniforms(seed, env_ids, tick):
    """12 U[0,1) draws per env for reset_idx: pos(3) euler(3) linvel(3) angvel(3)."""
    raw = raw_blocks(seed, env_ids, tick, STREAM_RESET, 3)
    return u32_to_unit_float(raw)  # [N,12]


def normals(seed, env_ids, tick, stream, count):
    """`count` N(0,1) draws per env via Box-Muller on consecutive u32 pairs.

    pair k uses raw[2k] (radius, open interval) and raw[2k+1] (angle); it yields
    z[2k] = r*cos(theta), z[2k+1] = r*sin(theta).
    """
    npairs = (count + 1) // 2
    nblocks = (2 * npairs + 3) // 4
    raw = raw_blocks(seed, env_ids, tick, stream, nblocks)
    u1 = u32_to_open_unit_float(raw[:, 0:2 * npairs:2])
    u2 = u32_to_unit_float(raw[:, 1:2 * npairs:2])
    r = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
    th = (TWO_PI_F32 * u2).astype(np.float32)
    z = np.empty((raw.shape[0], 2 * npairs), dtype=np.float32)
    z[:, 0::2] = r * np.cos(th)
    z[:, 1::2] = r * np.sin(th)
    return z[:, :count]
