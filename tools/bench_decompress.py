#!/usr/bin/env python3
"""Compressed column frames decoded in HBM (csrc/compress_kernels.hip, one wavefront per frame) against CPU LZ4 decoders on the same
box.  Compressed input resident in HBM when the timed region starts; the PCIe-inclusive figure (upload of the compressed bytes +
decode) is reported beside it.  usage: bench_decompress.py [rows]  -> one JSON object"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pyarrow as pa
import torch

import clickhouse_amd as ch
from clickhouse_amd import compression as CC
from oracle import compression as OC

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
codecs_only = len(sys.argv) > 2 and sys.argv[2] == "codecs"   # only the column codecs, at the full row count
ctx = ch.Context(0)
rng = np.random.Generator(np.random.PCG64(4))
res = []
cases = [("Int64 uniform in [0, 2^31) (C2's column)", rng.integers(0, 2**31, size=rows, dtype=np.int64)),
         ("Int64 ascending ids, small steps", np.cumsum(rng.integers(0, 4, size=rows)).astype(np.int64)),
         ("UInt8 discount 0..10", rng.integers(0, 11, size=rows * 4).astype(np.uint8))]
for name, arr, method in ([] if codecs_only else [(n_, a_, OC.METHOD_LZ4) for n_, a_ in cases] + [("Int64 ascending ids, CODEC(Delta(8), LZ4)", cases[1][1], OC.DELTA_LZ4)]):
    raw = arr.tobytes()
    for bs in (65536, 1 << 20):
        buf = OC.write_frames(raw, bs, method)
        frames = CC.parse_frames(buf)
        host = np.frombuffer(buf, dtype=np.uint8)
        up = ctx.upload(host)
        best = None
        for _ in range(5):
            ctx.synchronize()
            t0 = time.perf_counter()
            out = CC.decompress_frames(ctx, up, frames)
            ctx.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        assert out.numpy().tobytes() == raw
        del out
        t0 = time.perf_counter(); up2 = ctx.upload(host); o2 = CC.decompress_frames(ctx, up2, frames); ctx.synchronize(); t_pcie = time.perf_counter() - t0
        del up2, o2
        # CPU: Arrow's liblz4 (the library the reference links) and the plain-C restatement, one thread, first 64 frames
        sample = frames[:64]
        t0 = time.perf_counter()
        for m, off, size, dsize, post, stage in sample:
            pa.decompress(buf[off:off + size], stage, codec="lz4_raw", asbytes=True)
        t_arrow = time.perf_counter() - t0
        t0 = time.perf_counter()
        for m, off, size, dsize, post, stage in sample:
            OC.lz4_decompress(buf[off:off + size], stage)
        t_port = time.perf_counter() - t0
        sbytes = sum(f[3] for f in sample)
        res.append({"case": name, "frame_bytes": bs, "frames": len(frames), "raw_bytes": len(raw), "compressed_bytes": len(buf),
                    "ratio": len(raw) / len(buf), "gpu_ms": best * 1e3, "gpu_out_GBps": len(raw) / best / 1e9,
                    "algorithmic_GBps_in_plus_out": (len(raw) + len(buf)) / best / 1e9, "roofline_frac": (len(raw) + len(buf)) / best / 8e12,
                    "pcie_inclusive_out_GBps": len(raw) / t_pcie / 1e9, "cpu_liblz4_1thread_out_GBps": sbytes / t_arrow / 1e9,
                    "cpu_port_1thread_out_GBps": sbytes / t_port / 1e9})
        del up
        ctx.trim()
# round 3: the column codecs (one lane per frame for the bit-stream codecs DoubleDelta / Gorilla, one wave per 64-value block for T64), alone
# and as the second stage behind LZ4 (Multiple frames); 8192-value frames (the granule a MergeTree part writes)
crows = rows if codecs_only else min(rows, 20_000_000)
ts = np.cumsum(rng.integers(1, 20, size=crows)).astype(np.int64)                 # timestamps: DoubleDelta's case
gauge = (np.cumsum(rng.normal(size=crows)) * 0.25).round(2)                     # a slowly moving Float64 gauge: Gorilla's case
small = rng.integers(0, 5000, size=crows).astype(np.int64)                     # small integers in a wide type: T64's case
codec_cases = [("DoubleDelta, Int64 timestamps", ts, OC.METHOD_DOUBLE_DELTA, None), ("Gorilla, Float64 gauge", gauge, OC.METHOD_GORILLA, None),
               ("T64, Int64 values < 5000", small, OC.METHOD_T64, None), ("CODEC(DoubleDelta, LZ4)", ts, OC.METHOD_DOUBLE_DELTA, OC.METHOD_LZ4),
               ("CODEC(T64, LZ4)", small, OC.METHOD_T64, OC.METHOD_LZ4), ("CODEC(Gorilla, LZ4)", gauge, OC.METHOD_GORILLA, OC.METHOD_LZ4)]
if codecs_only:
    codec_cases = codec_cases[:3]
for name, arr, codec, general in codec_cases:
    raw = arr.tobytes()
    buf = OC.write_codec_frames(arr, codec, block_rows=8192) if general is None else OC.write_multiple_frames(arr, codec, general, block_rows=8192)
    best = None
    host = np.frombuffer(buf, dtype=np.uint8)
    for _ in range(4):
        ctx.synchronize()
        t0 = time.perf_counter()
        col = CC.read_column_file(ctx, host, arr.dtype, verify_checksums=False)   # walk + upload + decode (the upload is part of this figure)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    assert col.numpy().tobytes() == raw
    del col
    res.append({"case": name, "frame_values": 8192, "raw_bytes": len(raw), "compressed_bytes": len(buf), "ratio": len(raw) / len(buf),
                "walk_upload_decode_ms": best * 1e3, "out_GBps_pcie_inclusive": len(raw) / best / 1e9})
    ctx.trim()
print(json.dumps({"rows": rows, "results": res}))
