// lumahip_launch.hip -- launch geometry of the fused kernels: LDS bytes per workgroup, threads per workgroup, how many
// persistent workgroups.  Its own translation unit because these rules are PERFORMANCE-relevant the way the kernels are: the
// SHA that ties a committed rocprofv3 capture to the sources it ran on (lumahdrv_amd.capi.kernel_source_sha) covers the
// device code and this file, not the host-side plumbing of lumahip_core.hip / lumahip_host.hip.
#include "lumahip_internal.hpp"

using namespace lh;

namespace lhost {

size_t lds_bytes(const lumahip_ctx *c, bool encode_side, int cs_eff, bool ycode, bool half)
{
    const QuantDev &q = c->q;
    size_t b = 0;
    const size_t lut_b = ((size_t)(q.lut_len + q.pad) * 4 + 15) & ~(size_t)15;
    if (encode_side && ycode) {
        b += ((size_t)c->q_y.nbuckets * 4 + 15) & ~(size_t)15;
    } else if (encode_side) {
        if (q.mode == LUT_LITERAL_LDS)
            b += lut_b;
        if (q.mode == LUT_THRESH_LDS)
            b += ((size_t)q.nbuckets * 4 + 15) & ~(size_t)15;
        if (q.mode == LUT_LINKEY_LDS)
            b += ((size_t)q.nbuckets * 8 + 15) & ~(size_t)15;
    } else if (c->lut_in_lds) {
        b += lut_b;
        if (cs_eff == CS_LUV)
            b += (((size_t)q.maxC + 1) * 4 + 15) & ~(size_t)15;  // u'v' table of the Lu'v' decode kernels
        if (cs_eff == CS_YCBCR && q.ytab)
            b += lut_b + 2 * ((((size_t)q.maxC + 1) * 4 + 15) & ~(size_t)15);   // y table + the two chroma-term tables of the YCbCr decode kernels
    }
    if (encode_side && ycode && half)   // the half-input kernels: the table, and the small powf tables for their general path
        return b + (size_t)lds_half_bytes() + sizeof(PowfTables);
    if (cs_eff == CS_YCBCR)
        b += sizeof(PowfTablesWide);
    return b;
}

// Workgroup size of the fused kernels: 256 threads unless LUMAHIP_BLOCK says otherwise; search tables beyond
// 32 KiB per workgroup would leave too few waves per CU at that size (160 KiB of LDS per CU), so the workgroup
// grows with the table.
//   - `few_waves` (the encode kernels of the HBM-bound colour spaces on long launches, see grid_for): three 256-thread
//     workgroups per CU are the fastest configuration measured, so the workgroup stays at 256 threads as long as three
//     copies of the table fit the CU's LDS (LOG-12's 42 KiB of records: 3.9 % faster than four 512-thread workgroups).
int block_threads_for(const lumahip_ctx *c, size_t lds, bool few_waves, bool valu_bound)
{
    if (c->block_forced)
        return c->block_threads;
    // The per-pixel YCbCr kernels hold ~128 VGPRs: four waves per SIMD, i.e. TWO 512-thread workgroups per CU, are resident
    // whatever the LDS allows, so their workgroup stays at 512 threads as long as two copies of the tables fit (round 6: the
    // folded-powf tables took the decode kernels from 49 to 55 KiB; 1024-thread workgroups would only coarsen the shares).
    if (valu_bound && lds > 32 * 1024 && 2 * lds <= LUMAHIP_LDS_PER_WORKGROUP)
        return 512;
    if (few_waves && c->block_threads == 256 && 3 * lds <= LUMAHIP_LDS_PER_WORKGROUP)
        return 256;
    if (lds > 53 * 1024)
        return 1024;
    if (lds > 32 * 1024)
        return 512;
    return c->block_threads;
}

// Persistent workgroups: how many of them.  dir 0 = encode, 1 = decode.  The default is 2048 threads' worth per CU (8
// workgroups of 256), i.e. MORE than are resident at once for most kernels: the surplus is dispatched as resident ones
// retire, which evens out the tail of short launches.  The rules below were found on 20 x 3840x2160 launches by running both
// settings in one process (tools/bench/ab_inproc.py, profiles/r02_grid_sweep.txt) and then re-derived over {720p, 1080p, 4K, 8K} x
// {1, 2, 4, 8, 20, 50 frames} x {Lu'v', YCbCr} x {encode, decode} with every setting interleaved in one process
// (tools/bench/launch_rules_sweep.py, profiles/r03_launch_rules.txt: before / after tables).  lumahip_tune "grid_enc" / "grid_dec"
// (absolute) and "blocks_per_cu" (per CU, both directions) are measurement overrides.
int grid_for(const lumahip_ctx *c, int threads, int total_tiles, int dir, int few_writers, int ycbcr)
{
    int per_cu = c->blocks_per_cu > 0 ? c->blocks_per_cu : 2048 / threads;
    const bool rule = c->blocks_per_cu == 0;
    // The 4:2:0 16-bit decode kernels write 12 of their 15 bytes per pixel, and fewer concurrent writers suit the memory
    // system better than the default 8 workgroups of 256 threads per CU: 5 per CU on long launches (20 x 4K: 455 us against
    // 483; 8K x4 and longer likewise), 6 per CU on medium ones (1080p x8 ... x50, 4K x2 ... x8, 8K x1 ... x2: 2-5 % faster
    // than either 5 or 8), the default below that, where the finer tail matters more.  (The same change is 4 % SLOWER for
    // 4:4:4 Lu'v' and 10 % slower for the 8-bit profiles, so it is theirs only.)
    if (few_writers && rule && threads == 256) {
        if (total_tiles >= 60000)
            per_cu = 5;
        else if (total_tiles >= 6000)
            per_cu = 6;
        // ... unless the three colour planes are separate buffers (few_writers == 2: R, G, B spread over the HBM region
        // groups): then 6 per CU stays best on long launches too -- 20 x 4K, ordered: 0.417 ms against 0.436 with 5 and 0.432 with 8
        // (and against 0.454 / 0.470 / 0.482 for the packed layout; profiles/r03_striped_spread.txt).  That table also explains
        // round 2's "+11 % on one box, +1.2 % on another" for the striped output: the first figure was taken at 8 per CU, before
        // the rule above existed (packed 0.482 -> striped 0.432), the second after it (0.454 -> 0.436, +4 %).
        if (few_writers == 2 && total_tiles >= 6000)
            per_cu = 6;
    }
    // The encode kernels with 256-thread workgroups (tables up to 32 KiB; not YCbCr, which is VALU-bound and wants the
    // waves) run fastest with 3 workgroups per CU -- on long launches 3-6 % faster than with 8 (every colour space /
    // profile variant, profiles/r02_grid_sweep.txt; 6 per CU is 9 % SLOWER, 4 about as good), and the sweep over shapes shows
    // the same from one 1080p frame upwards (4K x1: 27.5 us against 28.0 with 4 and 29.4 with 8; 4K x4: 90.5 / 93.8 / 97.1;
    // 720p frames: no difference).  Fewer resident waves draw less power at the package limit and keep fewer streams open
    // in the memory system.
    if (dir == 0 && rule && threads == 256 && !ycbcr)
        per_cu = 3;
    // The YCbCr kernels are VALU-bound and only three of their 512-thread workgroups (49 KiB of LDS each) are resident
    // per CU: many more, smaller static shares balance the CUs better than one share per resident workgroup.  12 per CU is
    // never slower than the default 4 and 3-8 % faster from a few 1080p frames upwards; the longest launches gain another
    // 1-2 % from 18 (encode from 18 4K frames on, decode from 40).
    if (ycbcr && rule && threads == 512)
        per_cu = total_tiles >= (dir == 0 ? 36000 : 80000) ? 18 : 12;
    // The half-input YCbCr encode kernels (ycbcr == 2) are HBM-bound again, and ONE of their 1024-thread workgroups (~140 KiB
    // of LDS) is resident per CU; every further one stages the 124 KiB table anew.  Short launches want exactly the resident
    // set (720p x4: 16.2 us against 19.6 with 2 or more per CU; 1080p x1: 15.1 / 17.4; 4K x1: 28.6 / 30.4), long ones many small
    // static shares as the other YCbCr kernels do (4K x20: 422 us with 12 per CU against 466 with 2; 8K x20: 1659 / 1790);
    // profiles/r04_half_table_grid.txt, tools/bench/half_grid_sweep.py.
    if (ycbcr == 2 && rule && threads == 1024)
        per_cu = total_tiles < 3000 ? 1 : total_tiles < 12000 ? 3 : 12;
    long g = (long)c->num_cu * per_cu;
    // Inside an unordered section every launch keeps the grid it would have alone: two lanes of 3 (encode) / 5 (decode)
    // workgroups per CU each measured best (profiles/r03_layout_lab.txt: encode 0.780 of the roofline against 0.751 ordered,
    // 0.777 with 2 or 4 per CU each, 0.764 with three lanes; decode 0.760 against 0.715 ordered, 0.730 with 3.3 per CU each).
    // lumahip_tune("lane_grid_enc" / "lane_grid_dec") overrides it for measurements.
    if (c->lanes_active > 1 && c->lane_grid[dir] > 0)
        g = c->lane_grid[dir];
    if (c->grid_override[dir] > 0)
        g = c->grid_override[dir];
    if (g > total_tiles)
        g = total_tiles;
    if (g < 1)
        g = 1;
    return (int)g;
}

}  // namespace lhost
