// Read side of a "tile-sorted" partition layout (ground truth for DESIGN.md §4.3): the partition pass writes every 12288-row tile
// sorted by partition (a pure streaming write), and the aggregate pass of partition p then gathers one short run (~48 rows: 192 B of
// UInt32 keys + 384 B of 8-byte words) from every tile.  This measures that gather: one workgroup per partition, each wave takes
// tiles round-robin, U runs in flight per wave.  Run starts are pseudo-random inside the tile (unaligned), run length is `run` rows.
// build: hipcc -O3 --offload-arch=gfx950 tools/gather_runs_bench.hip -o tools/gather_runs_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned int u32;

template <int U>
__global__ __launch_bounds__(1024) void k_gather(const u32 * __restrict__ keys, const u64 * __restrict__ words, u32 tiles, u32 tile_rows, u32 run, u32 P, u64 * out, int nt)
{
    const u32 p = blockIdx.x % P, part_of = blockIdx.x / P, parts = gridDim.x / P; // several workgroups may share a partition (tile ranges)
    const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u64 acc = 0;
    const u32 t_begin = (u32)((u64)tiles * part_of / parts), t_end = (u32)((u64)tiles * (part_of + 1) / parts);
    for (u32 t = t_begin + wave * U; t < t_end; t += 16 * U)
    {
        u32 k[U];
        u64 w[U];
#pragma unroll
        for (int u = 0; u < U; ++u)
        {
            const u32 tt = t + u < t_end ? t + u : t_end - 1;
            // run start: p * run jittered by a per-tile hash (unaligned, like real partition boundaries)
            const u32 jitter = ((tt * 2654435761u) >> 27) % 16;
            u32 start = p * run + jitter;
            if (start + 64 > tile_rows) start = tile_rows - 64;
            const u64 row = (u64)tt * tile_rows + start + lane;
            const bool on = lane < run;
            if (nt)
            {
                k[u] = on ? __builtin_nontemporal_load(keys + row) : 0;
                w[u] = on ? __builtin_nontemporal_load(words + row) : 0;
            }
            else
            {
                k[u] = on ? keys[row] : 0;
                w[u] = on ? words[row] : 0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
            acc += k[u] ^ w[u];
    }
    if (acc == 0x123456789abcull) out[0] = acc;
}

int main(int argc, char ** argv)
{
    const u32 tile_rows = 12288;
    const u64 rows = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000000ull;
    const u32 tiles = (u32)(rows / tile_rows);
    u32 * keys; u64 * words; u64 * out;
    if (hipMalloc(&keys, (u64)tiles * tile_rows * 4) != hipSuccess || hipMalloc(&words, (u64)tiles * tile_rows * 8) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(keys, 1, (u64)tiles * tile_rows * 4);
    hipMemset(words, 1, (u64)tiles * tile_rows * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const u32 P = 256;
    for (u32 run : {48u, 24u, 12u})
        for (u32 parts : {1u, 2u, 4u})
            for (int U : {1, 2, 4, 8})
            for (int nt = 0; nt < 2; ++nt)
            {
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep)
                {
                    hipEventRecord(a);
                    const u32 Pn = tile_rows / run < P ? tile_rows / run : P;   // all rows of every tile are read once when Pn * run == tile_rows
                    const u32 Pfull = tile_rows / run;
                    // cover the whole tile: Pfull partitions (256 for run 48)
                    (void)Pn;
                    if (U == 1) hipLaunchKernelGGL(k_gather<1>, dim3(Pfull * parts), dim3(1024), 0, 0, keys, words, tiles, tile_rows, run, Pfull, out, nt);
                    if (U == 2) hipLaunchKernelGGL(k_gather<2>, dim3(Pfull * parts), dim3(1024), 0, 0, keys, words, tiles, tile_rows, run, Pfull, out, nt);
                    if (U == 4) hipLaunchKernelGGL(k_gather<4>, dim3(Pfull * parts), dim3(1024), 0, 0, keys, words, tiles, tile_rows, run, Pfull, out, nt);
                    if (U == 8) hipLaunchKernelGGL(k_gather<8>, dim3(Pfull * parts), dim3(1024), 0, 0, keys, words, tiles, tile_rows, run, Pfull, out, nt);
                    hipEventRecord(b);
                    hipEventSynchronize(b);
                    float ms = 0; hipEventElapsedTime(&ms, a, b);
                    if (ms < best) best = ms;
                }
                const double bytes = (double)tiles * tile_rows * 12.0;
                printf("run %2u rows  wgs/partition %u  U=%d nt=%d %8.3f ms  %.2f TB/s of useful bytes\n", run, parts, U, nt, best, bytes / (best * 1e-3) / 1e12);
            }
    return 0;
}
