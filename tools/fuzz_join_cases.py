"""Replays tests/test_gpu_fuzz.py::test_fuzz_join_against_oracle for one seed with a line per case (to locate a slow or stuck case).
usage: fuzz_join_cases.py <seed> [gpu|oracle|both]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import clickhouse_amd as ch
import oracle as O
O.build()
seed = int(sys.argv[1]); which = sys.argv[2] if len(sys.argv) > 2 else "both"
ctx = ch.Context(0)
rng = np.random.Generator(np.random.PCG64(4242 + seed))
variants = [(ch.JOIN_INNER, ch.STRICT_ALL, {}), (ch.JOIN_LEFT, ch.STRICT_ALL, {}), (ch.JOIN_LEFT, ch.STRICT_ANY, {}),
            (ch.JOIN_LEFT, ch.STRICT_ANY, {"any_take_last_row": True}), (ch.JOIN_INNER, ch.STRICT_ANY, {}),
            (ch.JOIN_LEFT, ch.STRICT_SEMI, {}), (ch.JOIN_LEFT, ch.STRICT_ANTI, {})]
for case in range(40):
    kind, strict, kw = variants[rng.integers(0, len(variants))]
    key_space = int([3, 100, 5000, 10**6, 2**40][rng.integers(0, 5)])
    n_blocks = int(rng.integers(0, 4))
    g = ch.HashJoin(kind, strict, kw.get("any_take_last_row", False), ctx=ctx)
    o = O.HashJoin(kind, strict, kw.get("any_take_last_row", False))
    sizes = []
    for _ in range(n_blocks):
        rows = int(rng.integers(0, 30_000))
        keys = rng.integers(0, key_space, size=rows, dtype=np.uint64)
        nm = (rng.random(rows) < 0.05).astype(np.uint8) if rng.random() < 0.4 else None
        jm = (rng.random(rows) < 0.9).astype(np.uint8) if rng.random() < 0.3 else None
        g.add_block(keys, null_map=nm, join_mask=jm)
        o.add_block(keys, null_map=nm, join_mask=jm)
        sizes.append(rows)
    left = rng.integers(0, key_space, size=int(rng.integers(0, 60_000)), dtype=np.uint64)
    lnm = (rng.random(left.shape[0]) < 0.03).astype(np.uint8) if rng.random() < 0.5 else None
    mjb = int([0, 0, 50, 4000][rng.integers(0, 4)])
    if mjb == 50:
        left, lnm = left[:2000], (None if lnm is None else lnm[:2000])
    print(f"case {case}: kind={kind} strict={strict} kw={kw} key_space={key_space} blocks={sizes} left={left.shape[0]} max={mjb}", flush=True)
    pos, it = 0, 0
    t0 = time.time()
    while True:
        if which in ("gpu", "both"):
            gl, gb, gr, gc = g.joined_pairs(left[pos:], None if lnm is None else lnm[pos:], max_joined_block_rows=mjb)
        if which in ("oracle", "both"):
            ol, ob, orow, oc = o.joined_pairs(left[pos:], None if lnm is None else lnm[pos:], max_joined_block_rows=mjb)
        c = gc if which != "oracle" else oc
        it += 1
        pos += c
        if pos >= left.shape[0] or c == 0:
            break
    print(f"    {it} probe calls, {time.time() - t0:.2f} s", flush=True)
