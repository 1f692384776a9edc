// Device kernels of libstag_hip.so (gfx950 / MI355X only).  Included once by stag_hip.hip.
//
// Data layout in HBM
//   packed rows   uint32[n][W]   bit f of sequence i = one-hot feature f (word f>>5, bit f&31),
//                                W = ceil(L/32) rounded up to 4 words (16-byte rows)
//   row windows   uint32[n]      the 4-word-aligned range outside which a row equals the
//                                "all gaps" row R0 (bit 0 of every column set), see r0_word()
//   coefficient blocks           each child row's NON-ZERO coefficients, sorted by feature, cut
//                                into chunks of CHUNK_WORDS = 16 row words (512 features, 64 B of
//                                a row) and padded per (row, chunk) to blocks of 4 entries
//                                {byte offset in row, bit, fp64 w} = 64 B;
//                                blk_off[row * NC + chunk] indexes them
//   gap chains    double[n_nodes][NC+1] x 2  what R0 scores before / from a chunk on (gappre, gapsuf)
//   grouping      node[n], pos[n], tasks[], order[64 per task]: sequences sorted by current tree
//                                node, cut into tasks of <= 64 sequences of one node
//   level records {node, prob}[D][n]  what each level decided for each sequence
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace stagk {

// ------------------------------------------------------------------------------------------
// Bounds-checked build (-DSTAG_CHECKED; `make checked` -> lib/libstag_hip_checked.so, selected at run time with
// STAG_HIP_LIB).  Every global-memory access of the kernels is written with the macros below.  In the product
// build they are the plain access (RDC: through the constant address space, see as_const).  In the checked build
// the address is first looked up in g_check, the table of the device allocations the host side knows of (its own
// buffers and the caller's, stag_hip.hip keeps it current): an access outside all of them is NOT made -- a load
// gives zero, a store or DMA is dropped -- and the first such access leaves {count, line of this file, address,
// bytes} in g_violation, which the host reads back after every call and reports as STAG_E_INVALID.  CHECK_THAT
// does the same for a condition (loop bounds taken from memory, offsets into LDS).  Such a build cannot fault on
// an index of its own making, so a run of the differential fuzz under it answers "kernel indexing or not".
// ------------------------------------------------------------------------------------------
#ifdef STAG_CHECKED
constexpr int CHECK_MAX = 1024;
struct CheckTable {
    uint32_t n, pad;
    uint64_t lo[CHECK_MAX], hi[CHECK_MAX];
};
__device__ CheckTable g_check;
__device__ unsigned long long g_violation[4];
__device__ __attribute__((noinline)) void violation(int line, uint64_t addr, uint64_t bytes) {
    if (atomicAdd(&g_violation[0], 1ull) == 0ull) {
        g_violation[1] = (unsigned long long)line;
        g_violation[2] = addr;
        g_violation[3] = bytes;
    }
}
__device__ __attribute__((noinline)) bool chk(const void *p, size_t bytes, int line) {
    const uint64_t a = (uint64_t)(uintptr_t)p;
    const uint32_t n = g_check.n;
    for (uint32_t i = 0; i < n; ++i)
        if (a >= g_check.lo[i] && a + bytes <= g_check.hi[i]) return true;
    violation(line, a, bytes);
    return false;
}
template <typename T>
__device__ __forceinline__ T checked_load(const T *p, int line) {
    if (chk(p, sizeof(T), line)) return *p;
    T zero;
    __builtin_memset(&zero, 0, sizeof(T));
    return zero;
}
template <typename T, typename V>
__device__ __forceinline__ void checked_store(T *p, const V &v, int line) {
    if (chk(p, sizeof(T), line)) *p = v;
}
__device__ __forceinline__ bool check_that(bool ok, int line, uint64_t a, uint64_t b) {
    if (!ok) violation(-line, a, b);
    return ok;
}
#define RD(arr, idx) checked_load(&(arr)[idx], __LINE__)
#define RDC(arr, idx) checked_load(&(arr)[idx], __LINE__)
#define RDK(arr, idx) checked_load(&(arr)[idx], __LINE__)   /* arr is already a STAG_CONST pointer */
#define WR(arr, idx, v) checked_store(&(arr)[idx], (v), __LINE__)
#define GUARD(ptr, bytes) chk((ptr), (bytes), __LINE__)
#define CHECK_THAT(cond, a, b) check_that((cond), __LINE__, (uint64_t)(a), (uint64_t)(b))
#else
#define RD(arr, idx) ((arr)[idx])
#define RDC(arr, idx) (as_const(arr)[idx])
#define RDK(arr, idx) ((arr)[idx])
#define WR(arr, idx, v) ((arr)[idx] = (v))
#define GUARD(ptr, bytes) true
#define CHECK_THAT(cond, a, b) true
#endif

#ifndef STAG_ABL
#define STAG_ABL 0                // timing ablations of k_walk_level (tools/walk_ablations.sh); 0 = the product
#endif
constexpr int TILE = 64;          // sequences per walk task = lanes of a wave
#ifndef STAG_NW
#define STAG_NW 4
#endif
constexpr int NW = STAG_NW;             // waves per walk workgroup (all share the 64 staged rows)
constexpr int CHUNK_WORDS = 16;   // row words per coefficient chunk (= one 64-byte sector of a row)
constexpr int WCAP_CHUNKS = 4;    // chunks of a row staged in LDS at a time (80 words, 320 B per row)
constexpr int ROW_STRIDE = WCAP_CHUNKS * CHUNK_WORDS + 1;  // odd: lane l reads word w at l*ROW_STRIDE+w, conflict-free
constexpr int BLOCK_ENTRIES = 4;  // coefficient lists are padded to 4 entries per (row, chunk)
constexpr int MFMA_SLICES = 6;    // signed base-256 digits per coefficient in the matrix-core walk
constexpr double MFMA_MIN_DENSITY = 0.05;   // non-zero fraction of the children's rows from which k_walk_mfma is chosen
constexpr int MFMA_MAX_K = 16;    // children per node it handles: two per block of 16 columns, 8 blocks
static_assert(WCAP_CHUNKS * CHUNK_WORDS == 64, "a staged slice is one dword per lane and row: all 64 lanes store into a row of ROW_STRIDE words");

struct Entry {      // 16 bytes
    uint32_t off;   // byte offset of the feature's word inside the row
    uint32_t bit;   // feature & 31
    double w;       // 0.0 for padding entries
};

// Read-only tables are addressed through the constant address space: a wave-uniform address
// then compiles to s_load_* (one fetch per wave into SGPRs) instead of a 64-lane vector load.
#ifdef STAG_CHECKED
#define STAG_CONST          /* plain loads, so that they can be checked */
#else
#define STAG_CONST __attribute__((address_space(4)))
#endif
template <typename T>
__device__ __forceinline__ const T STAG_CONST *as_const(const T *p) {
    return (const T STAG_CONST *)(uintptr_t)p;
}

struct Task {       // 64 bytes: the gather walk reads the first 32 with one scalar load, the matrix-core walk both halves
    int32_t node, count;
    int32_t c0, k;                     // children of node: ids c0 .. c0+k-1, k >= 2 (other nodes get no task)
    // HINT: everything a walk kernel would otherwise have to look up -- in loads that depend on each other -- before
    // it can fetch coefficients, for the window [hA, hB) (chunks) of the task's first sequence.  The grouping kernel
    // looks it up, where a million threads hide those loads; the walk then issues all its fetches right after this
    // record arrives.  Tasks are mostly made of rows with the same window (amplicon reads); when the union of the
    // task's windows differs, the walk does the lookups itself.  (k_walk_level looks its coefficient ranges up
    // itself: resolving them here made the grouping slower and the walk no faster.)
    int32_t hA, hB;
    int32_t mfma, pad1;                // mfma: the node has matrix-core tables (k_walk_mfma's task)
    // k_walk_mfma: the node's digit matrix and the hint window's union features
    uint64_t dig_at;                   // byte offset in dig of column 0's digit of the window's first feature
    uint32_t U;                        // bytes per column
    uint32_t f0, nf;                   // first union feature of the window (index into pfeat) and their number
    uint32_t pad2;
    double qscale;
};
static_assert(sizeof(Task) == 64, "two aligned 32-byte scalar loads");
// order[]: a sequence id and its R0 window, so the walk needs one load, not two.  Task t owns the
// 64 slots order[64 t .. 64 t + 63] (the first `count` are filled): the walk can fetch its task
// record and its sequences at the same time instead of one after the other.
struct SeqRef {
    int32_t seq;
    uint32_t win;
};
// What one level of the walk decides for one sequence; lvl[depth][seq], one 16-byte store.
// node < 0: the walk of this sequence has ended before this level.
struct __attribute__((aligned(16))) LevelRec {
    int32_t node;
    int32_t pad;
    double prob;        // 2.0 = "single child" placeholder (classify.py:37)
};
// plain vector types: unlike HIP's uint4 class they can be loaded from the constant address space
using i32x8 = __attribute__((ext_vector_type(8))) int32_t;
using i32x16 = __attribute__((ext_vector_type(16))) int32_t;

// ------------------------------------------------------------------------------------------
// encode: stag/align.py:28-58 (+ the filter of :160).  One wave per aligned string.
// ------------------------------------------------------------------------------------------
// Inclusive prefix sum over the 64 lanes with DPP moves (no LDS traffic): Kogge-Stone inside
// each row of 16 lanes, then the row totals are carried with row_bcast:15 / row_bcast:31.
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
    (void)lane;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return v;
}

// Word w of the "all gaps" row R0: bit 0 ("others") of every alignment column set, i.e. the
// features f < L with f % 5 == 0.  Amplicon reads equal R0 outside the stretch of columns they
// cover, so the walk stages only each row's *window* = the 4-word-aligned range outside which
// the row equals R0 (win = lo4 | hi4 << 16, in units of 4 words; lo4 >= hi4: empty).
__device__ __forceinline__ uint32_t r0_word(int w, int L) {
    const int s = (5 - (2 * w) % 5) % 5;   // 32 = 2 (mod 5): first bit of word w that starts a column
    uint32_t v = (uint32_t)(0x1084210842108421ull << s);
    const int64_t nbits = (int64_t)L - 32ll * w;
    if (nbits <= 0) v = 0;
    else if (nbits < 32) v &= (1u << nbits) - 1u;
    return v;
}

// 0xff in every byte of h80 that has its top bit set (h80 holds nothing but top bits): (h80 >> 7) * 0xff without the
// quarter-rate 32-bit multiply
__device__ __forceinline__ uint32_t byte_mask(uint32_t h80) { return (h80 << 1) - (h80 >> 7); }

// keep[lo_cut * 17 + hi_cut]: byte mask (4 dwords) of the bytes [lo_cut, hi_cut) of a 16-byte vector --
// the part of a record's first / last vector that belongs to the record.
struct alignas(16) KeepLut {      // (read with 16-byte loads)
    uint32_t m[17 * 17][4];
};
constexpr KeepLut make_keep_lut() {
    KeepLut t{};
    for (int lo = 0; lo <= 16; ++lo)
        for (int hi = 0; hi <= 16; ++hi)
            for (int j = 0; j < 4; ++j) {
                const int a0 = lo - 4 * j < 0 ? 0 : lo - 4 * j > 4 ? 4 : lo - 4 * j;
                const int b0 = hi - 4 * j < 0 ? 0 : hi - 4 * j > 4 ? 4 : hi - 4 * j;
                const uint32_t below_b = b0 >= 4 ? 0xffffffffu : (1u << (8 * b0)) - 1u;
                const uint32_t below_a = a0 >= 4 ? 0xffffffffu : (1u << (8 * a0)) - 1u;
                t.m[lo * 17 + hi][j] = below_b & ~below_a;
            }
    return t;
}
__device__ const KeepLut KEEP_LUT = make_keep_lut();

// Reductions inside each record's group of G lanes with DPP moves; the result is valid in the
// group's LAST lane (gl == G - 1).
template <int G, typename F>
__device__ __forceinline__ int group_reduce_last(int v, const int identity, F f) {
    v = f(v, __builtin_amdgcn_update_dpp(identity, v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = f(v, __builtin_amdgcn_update_dpp(identity, v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = f(v, __builtin_amdgcn_update_dpp(identity, v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = f(v, __builtin_amdgcn_update_dpp(identity, v, 0x118, 0xf, 0xf, false));  // row_shr:8
    if (G >= 32) v = f(v, __builtin_amdgcn_update_dpp(identity, v, 0x142, 0xa, 0xf, false));  // row_bcast:15
    if (G >= 64) v = f(v, __builtin_amdgcn_update_dpp(identity, v, 0x143, 0xc, 0xf, false));  // row_bcast:31
    return v;
}

// The record's LDS row holds the record's difference from R0 (every column a gap).  A lane XORs in, for each
// of its k columns, (one-hot group ^ gap group) at column running + before: zero for a gap, so lanes and
// steps that only see gaps touch nothing.  lo: bits 0..63, hi: bits 64..79 of the k groups.
// w_lo / w_hi track the words this lane changed (the row's window comes from them).
__device__ __forceinline__ void emit_groups(uint32_t *row, int W, int col, uint64_t lo, uint32_t hi, int &w_lo, int &w_hi) {
    if ((lo | hi) == 0ull) return;
    const int64_t bitoff = 5ll * col;
    const int word0 = (int)(bitoff >> 5);
    const int sh = (int)(bitoff & 31);
    const uint64_t a = lo << sh;
    const uint64_t carry = sh ? lo >> (64 - sh) : 0ull;
    const uint64_t bb = ((uint64_t)hi << sh) | carry;
    const uint32_t v4[4] = {(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)bb, (uint32_t)(bb >> 32)};
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (v4[i] && word0 + i < W) {
            atomicXor(&row[word0 + i], v4[i]);
            w_lo = min(w_lo, word0 + i);
            w_hi = max(w_hi, word0 + i + 1);
        }
}

// G = lanes per record (16, 32 or 64): a wave encodes 64/G records side by side, each lane
// taking 32 bytes (two 16-byte vectors) of its record per step.  The host picks G so that a record's length
// wastes few lane slots (a 1,501-byte record is 3 steps of 16 lanes = 1,536 bytes' worth).
// FULL: every word of the row is stored (stag_encode_a2m*, -S).  !FULL (the classify paths): only the 16-word
// chunks that hold the row's window are stored -- everything else equals R0, which the walk never reads from
// memory (k_walk_level scores a row outside its own window from the gap chains) -- so an amplicon read costs
// 192 B of row traffic instead of 944.  The words of such a row outside its window chunks keep whatever the
// buffer held before.
template <int G, bool FULL>
__global__ __launch_bounds__(256) void k_encode_a2m(const uint8_t *__restrict__ text, int64_t text_bytes,
                                                    const int64_t *__restrict__ offs, int64_t n, int L, int W,
                                                    double min_perc, uint32_t *__restrict__ rows,
                                                    int32_t *__restrict__ n_char_out,
                                                    int32_t *__restrict__ n_aligned_out,
                                                    int32_t *__restrict__ status_out,
                                                    uint32_t *__restrict__ win_out,
                                                    const int64_t *__restrict__ ends,
                                                    const int32_t *__restrict__ pre_status,
                                                    const int32_t *__restrict__ lead_trail) {
    // ends (may be null): record i is text[offs[i] .. ends[i]) instead of [offs[i] .. offs[i+1]);
    // pre_status (may be null): a non-zero entry is the record's final status (back-translation failed);
    // lead_trail (may be null): record i's text is its aligned string WITHOUT a run of lead_trail[2 i] '-' columns
    // at its start and lead_trail[2 i + 1] at its end (the host scanner drops them: a gap column is what the row
    // already says, so it only moves the column count)
    extern __shared__ uint32_t s_enc[];  // one W-word row per record of the workgroup
    constexpr int RPW = 64 / G;          // records per wave
    const int lane = threadIdx.x & 63;
    const int gl = lane & (G - 1);       // lane inside the record's group
    const int grp = lane / G;
    const int wave = threadIdx.x >> 6;
    const int64_t seq = ((int64_t)blockIdx.x * 4 + wave) * RPW + grp;
    const bool live = seq < n;
    uint32_t *row = s_enc + (size_t)(wave * RPW + grp) * W;
    // the LDS row collects what the record CHANGES relative to R0 (all gaps): it starts as zeros and is XOR-ed
    // with R0 on its way out (W is a multiple of 4)
    for (int w = gl * 4; w < W; w += 4 * G) *reinterpret_cast<uint4 *>(row + w) = make_uint4(0u, 0u, 0u, 0u);
    int w_lo = 0x7fffffff, w_hi = 0;     // words this lane changes

    // a record is clamped to the text buffer once, here: the loads below then need no bounds test of their own (a
    // 16-byte vector that starts before the record's end ends inside the buffer, whose size is a multiple of 16)
    int64_t begin = live ? RD(offs, seq) : 0, end = live ? (ends ? RD(ends, seq) : RD(offs, seq + 1)) : 0;
    begin = max(begin, (int64_t)0);
    end = max(min(end, text_bytes), begin);
    const int64_t base = begin & ~(int64_t)15;
    int running = 0, aligned = 0, trail = 0;
    if (live && lead_trail) {
        running = max(RD(lead_trail, 2 * seq), 0);
        trail = max(RD(lead_trail, 2 * seq + 1), 0);
    }
    // Positions inside the loop are 32-bit and relative to the record's first byte (a record is
    // shorter than 2 GB); only the load address is 64-bit.
    const int len = (int)(end - begin);
    const int lead = (int)(begin - base);        // bytes of the first vector that precede the record
    // A lane takes VPL = 2 vectors of 16 bytes per step: half as many steps, lanes and (per record) prologue and
    // epilogue instructions as with one.  text_bytes is a multiple of 16 (checked by the host side), so a 16-byte
    // vector is either entirely inside the buffer or entirely outside the record.  A load also fetches the mask of
    // the vector's bytes that belong to the record: [lo_cut, hi_cut) of 16 (all ones except in a record's first and
    // last vector).
    constexpr int VPL = 2, STEP = 16 * VPL * G;  // bytes of a record per step
    // A vector is fetched only if it starts before the record's end: the same bytes, no others, that every
    // earlier form of this kernel has read (positions relative to the record's first byte; rel is a multiple of 16
    // from -lead, so the address is 16-byte aligned).
    const uint8_t *const src = text + begin;
    auto load16 = [&](int rel, uint32_t (&o)[4], uint32_t (&k)[4]) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (rel < len) v = RD(reinterpret_cast<const uint4 *>(src + rel), 0);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
        uint4 q = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
        if (rel < 0 || rel + 16 > len) {     // the record's first / last vector, or one past its end
            const int lo_cut = min(16, max(0, -rel));
            const int hi_cut = min(16, max(0, len - rel));
            q = *reinterpret_cast<const uint4 *>(KEEP_LUT.m[lo_cut * 17 + hi_cut]);
        }
        k[0] = q.x; k[1] = q.y; k[2] = q.z; k[3] = q.w;
    };
    // one vector: its bytes' (one-hot group ^ gap group) in code, the skipped bytes in skip80, A/C/G/T/U counted
    auto classify16 = [&](const uint32_t (&wv)[4], const uint32_t (&keep)[4], uint32_t (&code)[4], uint32_t (&skip80)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t x = (wv[j] & keep[j]) | (0x0a0a0a0au & ~keep[j]);   // not the record's -> '\n' = skipped
            const uint32_t t = x & 0x7f7f7f7fu;
            // skipped: 'a'..'z' (HMM insert states, align.py:47) and 0x0a..0x0d (line breaks)
            const uint32_t ge61 = t + 0x1f1f1f1fu, ge7b = t + 0x05050505u;
            const uint32_t ge0a = t + 0x76767676u, ge0e = t + 0x72727272u;
            skip80[j] = ((ge61 & ~ge7b) | (ge0a & ~ge0e)) & ~x & 0x80808080u;
            // A/C/G/T/U differ in their low 3 bits (1,3,7,4,5): two byte-LUTs indexed by them give
            // the character that index stands for and its one-hot group; any other byte is "others"
            const uint32_t sel = x & 0x07070707u;
            const uint32_t expc = __builtin_amdgcn_perm(0x47ff5554u, 0x43ff41ffu, sel);
            const uint32_t oneh = __builtin_amdgcn_perm(0x04000202u, 0x08001000u, sel);
            const uint32_t y = x ^ expc;
            const uint32_t eq80 = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y) & 0x80808080u;  // byte == expected
            const uint32_t eqm = byte_mask(eq80);
            code[j] = (oneh ^ 0x01010101u) & eqm;   // group ^ gap group: zero for anything that is not A/C/G/T/U
            aligned += __popc(eq80);
        }
    };
    // 16 columns' groups -> 80 bits (byte j is column j)
    auto pack_static = [&](const uint32_t (&code)[4], uint64_t &lo, uint32_t &hi) {
        uint32_t pk[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            pk[j] = (code[j] & 0x1fu) | ((code[j] >> 3) & 0x3e0u) | ((code[j] >> 6) & 0x7c00u) | ((code[j] >> 9) & 0xf8000u);
        lo = (uint64_t)pk[0] | ((uint64_t)pk[1] << 20) | ((uint64_t)pk[2] << 40) | ((uint64_t)pk[3] << 60);
        hi = pk[3] >> 4;
    };
    // the kept bytes' groups packed densely (a skipped byte contributes no group); returns how many
    auto pack_dense = [&](const uint32_t (&code)[4], const uint32_t (&skip80)[4], uint64_t &lo, uint32_t &hi) {
        uint32_t pk[4];
        int cnt[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint32_t c = code[j] & ~byte_mask(skip80[j]);
            // kept flags as bytes, their running sums by one 24-bit multiply (byte i of run = kept among bytes 0..i,
            // i < 3), and the three shift amounts 5 * run the same way (a shift only looks at its low 5 bits)
            const uint32_t kept = (~skip80[j] >> 7) & 0x01010101u;
            const uint32_t run = __umul24(kept, 0x010101u);
            const uint32_t sh = __umul24(run, 5u);
            pk[j] = (c & 0x1fu) | (((c >> 8) & 0x1fu) << (sh & 31u)) | (((c >> 16) & 0x1fu) << ((sh >> 8) & 31u)) |
                    ((c >> 24) << ((sh >> 16) & 31u));
            cnt[j] = (int)(((run >> 16) & 0xffu) + (kept >> 24));
        }
        const int sh1 = 5 * cnt[0], sh2 = sh1 + 5 * cnt[1], sh3 = sh2 + 5 * cnt[2];   // <= 20, 40, 60
        lo = (uint64_t)pk[0] | ((uint64_t)pk[1] << sh1) | ((uint64_t)pk[2] << sh2) | ((uint64_t)pk[3] << sh3);
        hi = pk[3] >> min(64 - sh3, 31);   // the part of pk[3] (20 bits) beyond bit 63
        return cnt[0] + cnt[1] + cnt[2] + cnt[3];
    };
    // one step on the vectors wv / keep (fetched during the step before); the next step's go to nxt / nxt_keep
    auto step = [&](const uint32_t (&wv)[VPL][4], const uint32_t (&keep)[VPL][4], uint32_t (&nxt)[VPL][4],
                    uint32_t (&nxt_keep)[VPL][4], const int rel0) {
        const int rel = rel0 + gl * (16 * VPL);  // of the lane's first vector; negative only in the first step
#pragma unroll
        for (int v = 0; v < VPL; ++v)            // the next vectors are in flight while these are encoded
            load16(rel + STEP + 16 * v, nxt[v], nxt_keep[v]);
        // Aligned amplicon reads are mostly runs of '-': when every valid byte of every lane is a
        // gap, the step's output is a prefix of a constant bit pattern (bit 0 of each column).
        uint32_t not_gap = 0u;
#pragma unroll
        for (int v = 0; v < VPL; ++v)
#pragma unroll
            for (int j = 0; j < 4; ++j) not_gap |= (wv[v][j] ^ 0x2d2d2d2du) & keep[v][j];
        if (__ballot(not_gap != 0u) == 0ull) {
            // the row already says "gap" everywhere: only the column count moves on, by the bytes
            // of this step that belong to the record
            running += max(0, min(len, rel0 + STEP) - max(0, rel0));
            return;
        }
        uint32_t code[VPL][4], skip80[VPL][4];
        uint32_t any_skip = 0u;
#pragma unroll
        for (int v = 0; v < VPL; ++v) {
            classify16(wv[v], keep[v], code[v], skip80[v]);
            any_skip |= skip80[v][0] | skip80[v][1] | skip80[v][2] | skip80[v][3];
        }
        uint64_t lo[VPL];
        uint32_t hi[VPL];
        int at[VPL], total;   // where each vector's groups go (in columns of the record), and the step's columns
        if (__ballot(any_skip != 0u) == 0ull) {
            // no insert / line break anywhere in this step: byte j is column j, positions are static
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                pack_static(code[v], lo[v], hi[v]);
                at[v] = running + 16 * VPL * gl + 16 * v;
            }
            total = STEP;
        } else {
            int cnt[VPL], k = 0;
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                cnt[v] = pack_dense(code[v], skip80[v], lo[v], hi[v]);
                k += cnt[v];
            }
            // inclusive prefix sum inside each record's lane group (DPP: rows of 16, then row carries)
            int s = k;
            s += __builtin_amdgcn_update_dpp(0, s, 0x111, 0xf, 0xf, false);  // row_shr:1
            s += __builtin_amdgcn_update_dpp(0, s, 0x112, 0xf, 0xf, false);  // row_shr:2
            s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xf, 0xf, false);  // row_shr:4
            s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xf, 0xf, false);  // row_shr:8
            if (G >= 32) s += __builtin_amdgcn_update_dpp(0, s, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
            if (G >= 64) s += __builtin_amdgcn_update_dpp(0, s, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
            int before = running + s - k;
#pragma unroll
            for (int v = 0; v < VPL; ++v) {
                at[v] = before;
                before += cnt[v];
            }
            total = __shfl(s, (lane & ~(G - 1)) + G - 1);   // the inclusive sum of the group's last lane
        }
#pragma unroll
        for (int v = 0; v < VPL; ++v) emit_groups(row, W, at[v], lo[v], hi[v], w_lo, w_hi);
        running += total;
    };
    // two register sets used in turn, so that no step starts by copying the prefetched vectors
    uint32_t wa[VPL][4], ka[VPL][4], wb[VPL][4], kb[VPL][4];
#pragma unroll
    for (int v = 0; v < VPL; ++v) load16(gl * (16 * VPL) + 16 * v - lead, wa[v], ka[v]);
    for (int rel0 = -lead;;) {
        if (__ballot(rel0 < len) == 0ull) break;
        step(wa, ka, wb, kb, rel0);
        rel0 += STEP;
        if (__ballot(rel0 < len) == 0ull) break;
        step(wb, kb, wa, ka, rel0);
        rel0 += STEP;
    }
    running += trail;
    // totals over the record's lanes: valid in the group's last lane, which writes the record's scalars
    aligned = group_reduce_last<G>(aligned, 0, [](int x, int y) { return x + y; });
    uint32_t *dst = rows + (size_t)(live ? seq : 0) * W;
    // the window = the words some lane of the record changed (everything else still equals R0)
    w_lo = group_reduce_last<G>(w_lo, 0x7fffffff, [](int x, int y) { return min(x, y); });
    w_hi = group_reduce_last<G>(w_hi, 0, [](int x, int y) { return max(x, y); });
    int lo4 = 0xffff, hi4 = 0;
    if (w_hi > w_lo) { lo4 = w_lo >> 2; hi4 = (w_hi + 3) >> 2; }
    // a record with fewer columns than the database has no bits past its last column (it is
    // reported as LENGTH_MISMATCH and never classified; the row is still what align.py:37-58 gives)
    const int64_t valid_bits = 5ll * running;
    const bool short_row = valid_bits < (int64_t)L;
    if (short_row) { lo4 = 0; hi4 = W >> 2; }
    // words to store: all of them, or the 16-word chunks around the window (what the walk stages)
    int st_lo = 0, st_hi = W;
    if (!FULL) {
        const int last = (lane & ~(G - 1)) + G - 1;      // the group's last lane holds the reduced window
        const int l4 = __shfl(lo4, last), h4 = __shfl(hi4, last);
        const bool sr = __shfl((int)short_row, last) != 0;
        st_lo = (l4 * 4) & ~(CHUNK_WORDS - 1);
        st_hi = min(W, (h4 * 4 + CHUNK_WORDS - 1) & ~(CHUNK_WORDS - 1));
        if (h4 <= l4 || sr) st_hi = st_lo = 0;           // equals R0 everywhere / never classified: nothing to store
    }
    for (int w = st_lo + gl * 4; w < st_hi; w += 4 * G) {
        if (!live) break;
        uint32_t v[4] = {row[w], row[w + 1], row[w + 2], row[w + 3]};
        if ((int64_t)(w + 4) * 32 <= L) {
            const int s5 = (5 - (2 * w) % 5) % 5;
            const uint64_t r01 = 0x1084210842108421ull << s5, r23 = 0x1084210842108421ull << (s5 == 4 ? 0 : s5 + 1);
            v[0] ^= (uint32_t)r01; v[1] ^= (uint32_t)(r01 >> 32); v[2] ^= (uint32_t)r23; v[3] ^= (uint32_t)(r23 >> 32);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] ^= r0_word(w + j, L);
        }
        if (FULL && short_row) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t nb = valid_bits - 32ll * (w + j);
                if (nb <= 0) v[j] = 0u;
                else if (nb < 32) v[j] &= (1u << nb) - 1u;
            }
        }
        WR(reinterpret_cast<uint4 *>(dst + w), 0, make_uint4(v[0], v[1], v[2], v[3]));
    }
    if (live && gl == G - 1) {
        if (win_out) WR(win_out, seq, (uint32_t)lo4 | ((uint32_t)hi4 << 16));
        int st = 0;  // STAG_SEQ_OK
        if (running == 0) st = 5;
        // min_perc <= 0 (the default of `stag classify`): aligned / running * 100 >= 0 holds for every record
        // (a NaN threshold still fails every record)
        else if (!(min_perc <= 0.0) && !((double)aligned / (double)running * 100.0 >= min_perc)) st = 6;
        else if ((int64_t)running * 5 != L) st = 7;
        if (pre_status && RD(pre_status, seq)) st = RD(pre_status, seq);
        WR(n_char_out, seq, running);
        WR(n_aligned_out, seq, aligned);
        WR(status_out, seq, st);
    }
}

// ------------------------------------------------------------------------------------------
// back-translation: stag/align.py:63-94 protein2gene_alignment.  One wave per (protein
// alignment, gene) pair, 64 alignment characters per step.  '-' writes "---", an upper-case
// residue the next three nucleotides of the gene, a lower-case residue (insert state) skips
// three nucleotides, anything else does nothing.  Record i's output starts at 3 * prot_offs[i]
// (an upper bound of what the records before it emit, so no scan over records is needed) and
// ends at out_end[i].  status 8 = the length check of align.py:66-74 failed (the reference
// returns None there and then fails in convert_alignment).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_backtranslate(const uint8_t *__restrict__ prot,
                                                       const int64_t *__restrict__ prot_offs,
                                                       const uint8_t *__restrict__ gene,
                                                       const int64_t *__restrict__ gene_offs, int64_t n,
                                                       int check_length, uint8_t *__restrict__ out,
                                                       int64_t *__restrict__ out_begin, int64_t *__restrict__ out_end,
                                                       int32_t *__restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int64_t rec = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rec >= n) return;
    const int64_t p0 = RD(prot_offs, rec), p1 = RD(prot_offs, rec + 1);
    const int64_t g0 = RD(gene_offs, rec), glen = RD(gene_offs, rec + 1) - g0;
    const int64_t o0 = 3 * p0;
    if (check_length) {
        int64_t residues = 0;   // every character that is not '-' (align.py:65)
        for (int64_t p = p0 + lane; p < p1; p += 64) residues += RD(prot, p) != '-';
        residues = wave_sum((int)residues);
        if (glen != 3 * residues && glen - 3 != 3 * residues) {
            if (lane == 0) { WR(out_begin, rec, o0); WR(out_end, rec, o0); WR(status, rec, 8); }
            return;
        }
    }
    int64_t consumed = 0, emitted = 0;   // residues that took a codon / characters written, before this step
    for (int64_t base = p0; base < p1; base += 64) {
        const int64_t p = base + lane;
        const uint8_t c = p < p1 ? RD(prot, p) : (uint8_t)'.';
        const bool dash = c == '-', upper = c >= 'A' && c <= 'Z', lower = c >= 'a' && c <= 'z';
        const int takes = (upper || lower) ? 1 : 0;
        const int t_incl = wave_incl_scan(takes, lane);
        const int64_t g = 3 * (consumed + t_incl - takes);           // this residue's codon in the gene
        int e = 0;
        if (dash) e = 3;
        else if (upper) e = (int)min((int64_t)3, max((int64_t)0, glen - g));   // a slice past the end is shorter
        const int e_incl = wave_incl_scan(e, lane);
        uint8_t *dst = out + o0 + emitted + (e_incl - e);
        for (int j = 0; j < e; ++j) WR(dst, j, dash ? (uint8_t)'-' : RD(gene, g0 + g + j));
        consumed += __shfl(t_incl, 63);
        emitted += __shfl(e_incl, 63);
    }
    if (lane == 0) { WR(out_begin, rec, o0); WR(out_end, rec, o0 + emitted); WR(status, rec, 0); }
}

// The same back-translation for the fused protein path (stag_classify_protein_a2m*), which owns its output buffer and
// so can lay it out for speed: record i's output starts at 4 * prot_offs[i] -- always dword-aligned.  k_backtranslate
// spends its time issuing instructions (two wave scans and a round of shuffles per 64 characters, one byte load /
// store per lane and character: about 900 wave instructions for a 350-residue pair, 1.83 ms per million pairs).
// Here a lane takes BT_CPL = 8 consecutive characters, so a step covers 512 of them -- a 350-residue alignment is ONE
// step, with one pair of scans -- and all global traffic is in dwords: the gene bytes the step consumes (at most 1,536)
// are staged into LDS by dword loads, every lane writes what its characters emit into an LDS tile, and the tile's whole
// dwords leave by dword stores; the 0..3 bytes left over stay at the head of the tile for the next step (the last ones of
// a record leave as bytes).  The length check of align.py:66-74 is made in the same pass when the alignment is one step
// long.  The same characters out, the same status 8.
constexpr int BT_CPL = 8;                          // characters per lane and step
constexpr int BT_STEP = 64 * BT_CPL;               // characters per step
constexpr int BT_TILE_WORDS = (3 * BT_STEP + 3) / 4 + 4;   // dwords of a step's gene bytes / output, plus alignment slack
constexpr int BT_GENE_DW = (BT_TILE_WORDS + 63) / 64;      // staged gene dwords per lane

__global__ __launch_bounds__(256, 8) void k_backtranslate_fast(const uint8_t *__restrict__ prot, int64_t prot_bytes,
                                                               const int64_t *__restrict__ prot_offs,
                                                               const uint8_t *__restrict__ gene,
                                                               const int64_t *__restrict__ gene_offs, int64_t n,
                                                               uint8_t *__restrict__ out,
                                                               int64_t *__restrict__ out_begin, int64_t *__restrict__ out_end,
                                                               int32_t *__restrict__ status) {
    __shared__ uint32_t s_bt[4][2][BT_TILE_WORDS];   // per wave: gene bytes of the step | output tile
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t rec = (int64_t)blockIdx.x * 4 + wave;
    if (rec >= n) return;
    uint32_t *const s_gene = s_bt[wave][0], *const s_out = s_bt[wave][1];
    uint8_t *const s_gene8 = reinterpret_cast<uint8_t *>(s_gene), *const s_out8 = reinterpret_cast<uint8_t *>(s_out);
    // everything below is addressed as (wave-uniform base) + (32-bit offset): a record is shorter than 2 GB
    const int64_t p0 = RDC(prot_offs, rec), p1 = RDC(prot_offs, rec + 1);
    const int64_t g0 = RDC(gene_offs, rec), glen64 = RDC(gene_offs, rec + 1) - g0;
    const int plen = (int)(p1 - p0), glen = (int)glen64;
    const int tail_ok = (int)min((int64_t)plen, prot_bytes - p0);      // characters that may be read in whole dwords
    const uint8_t *const rp = prot + p0, *const rg = gene + g0;
    const int64_t o0 = 4 * p0;                    // dword-aligned whenever `out` is
    uint8_t *const ro = out + o0;
    uint32_t *const ro32 = reinterpret_cast<uint32_t *>(ro);
    auto reject = [&]() {                         // align.py:66-74: the reference returns None here
        if (lane == 0) { WR(out_begin, rec, o0); WR(out_end, rec, o0); WR(status, rec, 8); }
    };
    // dword d of this lane's characters from record position q on ('.', a character that does nothing, past the end)
    auto load_dword = [&](int q) {
        uint32_t w = 0x2e2e2e2eu;
        if (q + 4 <= tail_ok) {
            w = RD(reinterpret_cast<const uint32_t *>(rp + q), 0);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (q + j < plen) w = (w & ~(0xffu << (8 * j))) | ((uint32_t)RD(rp, q + j) << (8 * j));
        }
        return w;
    };
    // per dword: byte masks (0x80 per byte) of '-', 'A'..'Z', 'a'..'z'
    auto classes = [](uint32_t w, uint32_t &dash, uint32_t &upper, uint32_t &lower) {
        const uint32_t t = w & 0x7f7f7f7fu, hi = w & 0x80808080u;
        const uint32_t y = w ^ 0x2d2d2d2du;
        dash = ~(((y & 0x7f7f7f7fu) + 0x7f7f7f7fu) | y) & 0x80808080u;
        upper = ((t + 0x3f3f3f3fu) & ~(t + 0x25252525u)) & ~hi & 0x80808080u;     // 0x41 <= t <= 0x5a
        lower = ((t + 0x1f1f1f1fu) & ~(t + 0x05050505u)) & ~hi & 0x80808080u;     // 0x61 <= t <= 0x7a
    };
    const bool one_step = plen <= BT_STEP;
    if (!one_step) {        // a long alignment: the length check needs every character before the first is written
        int nondash = 0;
        for (int q = lane * BT_CPL; q < plen; q += BT_STEP)
#pragma unroll
            for (int d = 0; d < BT_CPL / 4; ++d) {
                uint32_t dash, upper, lower;
                classes(load_dword(q + 4 * d), dash, upper, lower);
                nondash += min(4, max(0, plen - (q + 4 * d))) - __popc(dash);
            }
        const int64_t residues = wave_sum(nondash);
        if (glen64 != 3 * residues && glen64 - 3 != 3 * residues) { reject(); return; }
    }
    int consumed = 0, emitted = 0;                // residues that took a codon / characters written, before this step
    for (int base = 0; base < plen; base += BT_STEP) {
        // ---- the gene bytes this step can consume (at most 3 per character) are fetched first, in dwords that are
        // aligned in memory, so that they travel while the characters are read and counted
        const int G0 = 3 * consumed, gspan = min(glen - G0, 3 * BT_STEP);
        const int gshift = (int)(reinterpret_cast<uintptr_t>(rg + G0) & 3);
        const int nd = gspan > 0 ? (gshift + gspan + 3) >> 2 : 0;                  // <= BT_TILE_WORDS - 3
        const uint8_t *const gbase = rg + (G0 - gshift);                           // wave-uniform, dword-aligned
        uint32_t gv[BT_GENE_DW];
#pragma unroll
        for (int i = 0; i < BT_GENE_DW; ++i) {
            const int d = lane + 64 * i;
            gv[i] = 0;
            if (d < nd) {
                if (4 * d + 4 - gshift <= glen - G0) {
                    gv[i] = RD(reinterpret_cast<const uint32_t *>(gbase), d);
                } else {                                            // the record's last dword: only its own bytes
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int off = 4 * d + j - gshift;         // relative to G0
                        if (off >= -G0 && off < glen - G0) gv[i] |= (uint32_t)RD(gbase, 4 * d + j) << (8 * j);
                    }
                }
            }
        }
        const int q = base + lane * BT_CPL;
        uint32_t w[BT_CPL / 4], dash[BT_CPL / 4], upper[BT_CPL / 4], lower[BT_CPL / 4];
        int takes = 0, e = 0, nondash = 0;
#pragma unroll
        for (int d = 0; d < BT_CPL / 4; ++d) {
            w[d] = load_dword(q + 4 * d);
            classes(w[d], dash[d], upper[d], lower[d]);
            takes += __popc(upper[d] | lower[d]);
            e += 3 * __popc(dash[d] | upper[d]);
            nondash += min(4, max(0, plen - (q + 4 * d))) - __popc(dash[d]);       // what align.py:65 counts
        }
        if (one_step) {
            const int64_t residues = wave_sum(nondash);
            if (glen64 != 3 * residues && glen64 - 3 != 3 * residues) { reject(); return; }
        }
        // (after the length check a codon is never cut short by the gene's end: 3 * letters <= 3 * residues <= glen)
        const int t_incl = wave_incl_scan(takes, lane), e_incl = wave_incl_scan(e, lane);
        const int t_total = __shfl(t_incl, 63), e_total = __shfl(e_incl, 63);
        // ---- the staged gene bytes -> LDS
#pragma unroll
        for (int i = 0; i < BT_GENE_DW; ++i)
            if (lane + 64 * i < nd) s_gene[lane + 64 * i] = gv[i];
        // ---- this lane's characters -> the output tile, behind the bytes the last step left over
        const int oshift = emitted & 3;
        int at = oshift + e_incl - e;                                       // in the tile
        int from = 3 * (t_incl - takes) + gshift;                           // in the staged gene bytes
#pragma unroll
        for (int d = 0; d < BT_CPL / 4; ++d)
#pragma unroll 1
            for (int j = 0; j < 4; ++j) {
                const uint32_t bit = 0x80u << (8 * j);
                const bool is_dash = dash[d] & bit, is_upper = upper[d] & bit, is_lower = lower[d] & bit;
                if (is_dash || is_upper) {
                    for (int b = 0; b < 3; ++b) s_out8[at + b] = is_dash ? (uint8_t)'-' : s_gene8[from + b];
                    at += 3;
                }
                from += (is_upper || is_lower) ? 3 : 0;
            }
        // ---- whole dwords of the tile -> memory; what is left over moves to the tile's head
        const int total = oshift + e_total, nd_out = total >> 2;
        for (int d = lane; d < nd_out; d += 64) WR(ro32, ((emitted - oshift) >> 2) + d, s_out[d]);
        const uint32_t left = s_out[nd_out];
        if (lane == 0) s_out[0] = left;
        consumed += t_total;
        emitted += e_total;
    }
    const int rest = emitted & 3;
    if (lane < rest) WR(ro, (emitted - rest) + lane, s_out8[lane]);
    if (lane == 0) { WR(out_begin, rec, o0); WR(out_end, rec, o0 + emitted); WR(status, rec, 0); }
}

// ------------------------------------------------------------------------------------------
// -s lines: stag/classify.py:16-20 (alignment_reader) for lines in the form stag itself writes (-S,
// classify.py:160-162): after the id, fields that are the single characters '0' / '1', separated by single tabs.
// body = text[body_begin, body_end): the line after the id's tab, right-stripped (the host finds the lines; it does
// not look at the fields).  One wave per line; per step lane l takes 64 bytes = 32 fields = one word of the row.
//   n_fields = fields of the line (0 when the body is not of that form)
//   flag     = 1: the body is not "x<tab>x<tab>...x" with every x in {0, 1} -> the caller parses the line the
//              reference's way (int() per field) and supplies the row; 0: row written (all zero when n_fields != L,
//              such a row is never classified)
// The text buffer must be readable up to the next multiple of 4 after every body_end.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_parse_onehot(const uint8_t *__restrict__ text, int64_t text_bytes,
                                                      const int64_t *__restrict__ body_begin,
                                                      const int64_t *__restrict__ body_end, int64_t n, int L, int W,
                                                      uint32_t *__restrict__ rows, int32_t *__restrict__ n_fields,
                                                      int32_t *__restrict__ flag) {
    const int lane = threadIdx.x & 63;
    const int64_t line = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (line >= n) return;
    const int64_t b = RD(body_begin, line), e = RD(body_end, line);
    const int64_t blen = e - b;                  // "x\tx\t...x": 2 nf - 1 bytes
    const bool form = b >= 0 && e <= text_bytes && blen > 0 && (blen & 1) && (blen + 1) / 2 <= 0x7fffffff;
    const int64_t nf = form ? (blen + 1) / 2 : 0;
    const bool takes = form && nf == L;
    uint32_t *row = rows + (size_t)line * W;
    constexpr uint32_t PAT = 0x09300930u;        // '0' <tab> '0' <tab>, little endian
    uint32_t bad = 0;
    if (form) {
        const int64_t words = (blen + 64) / 64;  // words of 32 fields the body touches
        const int64_t w_end = takes && W > words ? (int64_t)W : words;   // (a row is padded to a multiple of 4 words)
        for (int64_t w0 = 0; w0 < w_end; w0 += 64) {
            const int64_t w = w0 + lane;
            const int64_t pos = b + 64 * w;      // first byte of this lane's 32 fields
            uint32_t word = 0;
            if (w < words) {
                const int64_t base = pos & ~(int64_t)3;
                const int sh = (int)(pos & 3);
                uint32_t d[17];
#pragma unroll
                for (int i = 0; i < 17; ++i) {   // aligned dwords that hold bytes of the body
                    const int64_t at = base + 4 * i;
                    d[i] = at < e ? RD(reinterpret_cast<const uint32_t *>(text + at), 0) : 0u;
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    // the 4 bytes at pos + 4 i; v_alignbyte: bytes sh .. sh + 3 of the pair
                    uint32_t x = __builtin_amdgcn_alignbyte(d[i + 1], d[i], (uint32_t)sh);
                    const int64_t left = e - (pos + 4 * i);          // bytes of the body from here on
                    if (left < 4) {              // past the body: as if the line went on with "0\t0\t..."
                        const uint32_t keep = left <= 0 ? 0u : (1u << (8 * (int)left)) - 1u;
                        x = (x & keep) | (PAT & ~keep);
                    }
                    bad |= (x & 0xfffefffeu) ^ PAT;
                    word |= ((x & 1u) | ((x >> 15) & 2u)) << (2 * i);
                }
            }
            if (takes && w < W) WR(row, w, word);
        }
    }
    const bool any_bad = __ballot(bad != 0u) != 0ull;
    if (!takes || any_bad)
        for (int w = lane; w < W; w += 64) WR(row, w, 0u);
    if (lane == 0) {
        WR(n_fields, line, (int32_t)nf);
        WR(flag, line, (!form || any_bad) ? 1 : 0);
    }
}

// find_n_aligned_characters (classify.py:79-87) and the R0 window for rows that did not come
// from k_encode_a2m.
__global__ __launch_bounds__(256) void k_count_aligned(const uint32_t *__restrict__ rows, int64_t n, int L, int W,
                                                       int32_t *__restrict__ n_aligned,
                                                       uint32_t *__restrict__ win_out) {
    const int lane = threadIdx.x & 63;
    const int64_t seq = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seq >= n) return;
    const uint32_t *row = rows + (size_t)seq * W;
    int gaps = 0, lo = 0x3ffff, hi = 0;
    for (int w0 = 0; w0 < W; w0 += 64) {
        const int w = w0 + lane;
        bool differs = false;
        if (w < W) {
            const uint32_t x = RD(row, w), r0 = r0_word(w, L);
            gaps += __popc(x & r0);
            differs = x != r0;
        }
        const unsigned long long d = __ballot(differs);
        if (d) {
            lo = min(lo, w0 + (int)__ffsll((long long)d) - 1);
            hi = max(hi, w0 + 64 - (int)__clzll((long long)d));
        }
    }
    gaps = wave_sum(gaps);
    if (lane == 0) {
        WR(n_aligned, seq, (L + 4) / 5 - gaps);
        if (win_out) WR(win_out, seq, hi > lo ? (uint32_t)(lo >> 2) | ((uint32_t)((hi + 3) >> 2) << 16) : 0xffffu);
    }
}

// ------------------------------------------------------------------------------------------
// concatenation of marker-gene rows: stag/classify_genome.py:364-388 on packed rows.  Output row
// j (one genome) is the genes' rows laid end to end at bit offsets goff[p]; a gene the genome
// lacks (src_row < 0) contributes zeros.  One thread per output word.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_concat_bits(int64_t n_genomes, int n_genes, const int64_t *__restrict__ goff,
                                                     const int64_t *__restrict__ src_base,
                                                     const int32_t *__restrict__ src_wpr,
                                                     const int32_t *__restrict__ src_row,
                                                     const uint32_t *__restrict__ arena, uint32_t *__restrict__ out,
                                                     int W_out) {
    const int w = blockIdx.y * 256 + threadIdx.x;   // genomes along x: a grid's x extent has no practical bound
    const int64_t j = blockIdx.x;
    if (w >= W_out || j >= n_genomes) return;
    const int64_t lo = 32ll * w, hi = lo + 32;
    int a = 0, b = n_genes;   // first gene whose bits end after lo
    while (a < b) {
        const int m = (a + b) >> 1;
        if (RD(goff, m + 1) > lo) b = m; else a = m + 1;
    }
    uint32_t v = 0;
    for (int p = a; p < n_genes && RD(goff, p) < hi; ++p) {
        const int r = RD(src_row, j * n_genes + p);
        if (r < 0) continue;
        const int64_t d0 = max(lo, RD(goff, p)), d1 = min(hi, RD(goff, p + 1));
        const int nb = (int)(d1 - d0);
        if (nb <= 0) continue;
        const int64_t s0 = d0 - RD(goff, p);
        const int wpr = RD(src_wpr, p);
        const uint32_t *row = arena + RD(src_base, p) + (int64_t)r * wpr;
        const int i = (int)(s0 >> 5), sh = (int)(s0 & 31);
        uint32_t x = RD(row, i) >> sh;
        if (sh && i + 1 < wpr) x |= RD(row, i + 1) << (32 - sh);
        if (nb < 32) x &= (1u << nb) - 1u;
        v |= x << (int)(d0 - lo);
    }
    WR(out, j * W_out + w, v);
}

// ------------------------------------------------------------------------------------------
// grouping by current node
// ------------------------------------------------------------------------------------------
// Only for a tree without levels (a root with no children): status is otherwise set by the
// first k_assign.
__global__ __launch_bounds__(256) void k_init_status(int64_t n, const int32_t *__restrict__ status_in,
                                                     int32_t *__restrict__ status) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s < n) WR(status, s, status_in ? RD(status_in, s) : 0);
}

// pos[s] = rank of s among the sequences of its node, count[node] = sequences at that node.
// A hot node (top levels: every sequence stands on one of a few nodes) would serialise
// returning atomics on one word (~88 per us), so sequences are first counted per workgroup:
//  * levels with <= ASSIGN_BINS nodes: LDS histogram over the level's node range, then one
//    global atomic per (workgroup, non-empty bin);
//  * wider levels: per-wave grouping by ballot (ALU only), then one atomic per (wave, node).
constexpr int ASSIGN_BINS = 15360;   // 60 KB of LDS counters
constexpr int ASSIGN_THREADS = 1024;
constexpr int ASSIGN_ITEMS = 4;      // sequences per thread: a workgroup counts 4,096 sequences, so a
                                     // hot node takes a quarter of the same-address global atomics
constexpr int ASSIGN_SPAN = ASSIGN_THREADS * ASSIGN_ITEMS;

//
// The node a sequence stands on comes from the previous level's record (level 0: the root for
// every sequence whose incoming status is OK; that status is copied to the output here).  A
// sequence that is not on a node of this level has ended its walk: its record of this level
// says so, which is what the next level and the rank cut read.
__global__ __launch_bounds__(ASSIGN_THREADS) void k_assign(int64_t n, int lo, int hi,
                                                           const LevelRec *__restrict__ prev,
                                                           LevelRec *__restrict__ cur,
                                                           const int32_t *__restrict__ status_in,
                                                           int32_t *__restrict__ status,
                                                           int32_t *__restrict__ node,
                                                           int32_t *__restrict__ count, int32_t *__restrict__ pos) {
    __shared__ int s_cnt[ASSIGN_BINS];
    const int lane = threadIdx.x & 63;
    const int64_t s0 = (int64_t)blockIdx.x * ASSIGN_SPAN + threadIdx.x;
    int nd[ASSIGN_ITEMS];
#pragma unroll
    for (int j = 0; j < ASSIGN_ITEMS; ++j) {
        const int64_t s = s0 + (int64_t)j * ASSIGN_THREADS;
        nd[j] = -1;
        if (s < n) {
            if (prev) {
                nd[j] = RD(&prev[s].node, 0);
            } else {
                const int st = status_in ? RD(status_in, s) : 0;
                WR(status, s, st);
                nd[j] = st == 0 ? 0 : -1;
            }
            if (nd[j] < lo || nd[j] >= hi) {
                nd[j] = -1;
                WR(&cur[s].node, 0, -1);
            }
            WR(node, s, nd[j]);
        }
    }
    const int nbins = hi - lo;
    int rank[ASSIGN_ITEMS];
    if (nbins <= ASSIGN_BINS) {
        for (int b = threadIdx.x; b < nbins; b += ASSIGN_THREADS) s_cnt[b] = 0;
        __syncthreads();
        if (nbins <= 32) {
            // a handful of nodes (top levels): thousands of LDS atomics on the same few words would
            // serialise; count per wave with ballots and add once per (wave, node)
#pragma unroll
            for (int j = 0; j < ASSIGN_ITEMS; ++j) rank[j] = 0;
            for (int b = 0; b < nbins; ++b) {
                unsigned long long m[ASSIGN_ITEMS];
                int tot = 0;
#pragma unroll
                for (int j = 0; j < ASSIGN_ITEMS; ++j) {
                    m[j] = __ballot(nd[j] == lo + b);
                    tot += __popcll(m[j]);
                }
                int base_b = 0;
                if (lane == 0 && tot) base_b = atomicAdd(&s_cnt[b], tot);
                base_b = __builtin_amdgcn_readfirstlane(base_b);
#pragma unroll
                for (int j = 0; j < ASSIGN_ITEMS; ++j) {
                    if (nd[j] == lo + b) rank[j] = base_b + __popcll(m[j] & ((1ull << lane) - 1ull));
                    base_b += __popcll(m[j]);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < ASSIGN_ITEMS; ++j) rank[j] = nd[j] >= 0 ? atomicAdd(&s_cnt[nd[j] - lo], 1) : 0;
        }
        __syncthreads();
        // one global atomic per non-empty bin; four bins per thread and step so that the returning
        // atomics are in flight together instead of one round trip after the other
        for (int b0 = threadIdx.x; b0 < nbins; b0 += 4 * ASSIGN_THREADS) {
            int c[4], r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = b0 + u * ASSIGN_THREADS;
                c[u] = b < nbins ? s_cnt[b] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                r[u] = c[u] && GUARD(&count[lo + b0 + u * ASSIGN_THREADS], 4) ? atomicAdd(&count[lo + b0 + u * ASSIGN_THREADS], c[u]) : 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = b0 + u * ASSIGN_THREADS;
                if (b < nbins) s_cnt[b] = r[u];
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < ASSIGN_ITEMS; ++j)
            if (nd[j] >= 0) WR(pos, s0 + (int64_t)j * ASSIGN_THREADS, s_cnt[nd[j] - lo] + rank[j]);
        return;
    }
#pragma unroll
    for (int j = 0; j < ASSIGN_ITEMS; ++j) {
        const int ndj = nd[j];
        bool todo = ndj >= 0;
        int rk = 0, cnt = 0, leader_lane = lane;
        for (;;) {
            const unsigned long long pending = __ballot(todo);
            if (!pending) break;
            const int leader = __ffsll((long long)pending) - 1;
            const int n0 = __shfl(ndj, leader);
            const bool mine = todo && ndj == n0;
            const unsigned long long same = __ballot(mine);
            if (mine) {
                rk = __popcll(same & ((1ull << lane) - 1ull));
                cnt = __popcll(same);
                leader_lane = leader;
                todo = false;
            }
        }
        int base = 0;
        if (ndj >= 0 && lane == leader_lane && GUARD(&count[ndj], 4)) base = atomicAdd(&count[ndj], cnt);
        base = __shfl(base, leader_lane);
        if (ndj >= 0) WR(pos, s0 + (int64_t)j * ASSIGN_THREADS, base + rk);
    }
}

// A node whose sequences need no scoring: a leaf (k == 0: the walk ends), a single child (k == 1: the
// placeholder 2 of classify.py:37, its classifier is never evaluated) or a node at which the reference
// raises (err).  k_scatter / k_root_level write those sequences' level records themselves; only the
// other nodes get walk tasks.
__device__ __forceinline__ bool node_is_trivial(const int32_t *__restrict__ child_first,
                                                const uint8_t *__restrict__ node_err, int nd) {
    return RD(child_first, nd + 1) - RD(child_first, nd) <= 1 || RD(node_err, nd) != 0;
}

// One workgroup: exclusive scan of the task counts over the nodes [lo, hi) of one level.
// Each thread owns a contiguous run of nodes; the 1024 run totals are scanned with DPP inside
// waves and one LDS exchange of the 16 wave totals.
__global__ __launch_bounds__(1024) void k_scan_level(int lo, int hi, const int32_t *__restrict__ count,
                                                     const int32_t *__restrict__ child_first,
                                                     const uint8_t *__restrict__ node_err,
                                                     int32_t *__restrict__ task_off,
                                                     int32_t *__restrict__ n_tasks) {
    __shared__ int s_wb[16];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (hi - lo + 1023) / 1024;
    const int i0 = lo + t * per, i1 = min(hi, i0 + per);
    auto tasks_of = [&](int i) { return node_is_trivial(child_first, node_err, i) ? 0 : (RD(count, i) + TILE - 1) / TILE; };
    int sb = 0;
    for (int i = i0; i < i1; ++i) sb += tasks_of(i);
    const int ib = wave_incl_scan(sb, lane);
    if (lane == 63) s_wb[wave] = ib;
    __syncthreads();
    int base_b = 0, tot_b = 0;
    for (int w = 0; w < 16; ++w) {
        if (w < wave) base_b += s_wb[w];
        tot_b += s_wb[w];
    }
    int rb = base_b + ib - sb;
    for (int i = i0; i < i1; ++i) {
        WR(task_off, i, rb);
        rb += tasks_of(i);
    }
    if (t == 0) WR(n_tasks, 0, tot_b);
}

// What a trivial node decides for every sequence standing on it (see node_is_trivial).
__device__ __forceinline__ void settle_trivial(int c0, int k, int err, int64_t s, LevelRec *__restrict__ cur,
                                               int32_t *__restrict__ status) {
    LevelRec r;
    r.pad = 0;
    if (err) {            // the reference raises here (classify.py:40 / :24-25)
        WR(status, s, err);
        r.node = -1; r.prob = 0.0;
    } else if (k == 1) {  // classify.py:36-37: the only child, placeholder score 2, classifier not evaluated
        r.node = c0; r.prob = 2.0;
    } else {              // a leaf: the walk ends (classify.py:34-35)
        r.node = -1; r.prob = 0.0;
    }
    WR(cur, s, r);
}

// Tables a grouping kernel needs to resolve a task's hints.
struct HintTables {
    const uint32_t *blk_off;    // gather walk: [n_nodes * NC + 1]
    const uint8_t *mfma_ok;     // [n_nodes] (null: no matrix-core tables)
    const uint32_t *pf_off;
    const uint64_t *dig_base;
    const uint32_t *dig_ld;
    const double *qscale;
    int NC;
};

// The task record of <= 64 sequences of node nd, with the hint (see Task) taken from one of its sequences' window.
__device__ __forceinline__ Task make_task(int nd, int count, int c0, int k, uint32_t win, const HintTables &h) {
    Task tk;
    tk.node = nd; tk.count = count; tk.c0 = c0; tk.k = k; tk.pad1 = 0; tk.pad2 = 0;
    const int l4 = (int)(win & 0xffffu), h4 = (int)(win >> 16);
    int hA = 0, hB = 0;
    if (h4 > l4) { hA = l4 >> 2; hB = min((h4 + 3) >> 2, h.NC); }
    tk.hA = hA; tk.hB = hB;
    const bool mfma = h.mfma_ok && RD(h.mfma_ok, nd);
    tk.mfma = mfma ? 1 : 0;
    tk.dig_at = 0; tk.U = 0; tk.f0 = 0; tk.nf = 0; tk.qscale = 0.0;
    if (mfma) {
        const uint32_t *po = h.pf_off + (size_t)nd * h.NC;
        const uint32_t first = RD(po, 0), f0 = RD(po, hA), f1 = RD(po, hB);
        tk.U = RD(h.dig_ld, nd);
        tk.f0 = f0;
        tk.nf = f1 - f0;
        tk.dig_at = RD(h.dig_base, nd) + (f0 - first);
        tk.qscale = RD(h.qscale, nd);
    }
    return tk;
}

__global__ __launch_bounds__(256) void k_scatter(int64_t n, int lo, int hi, const int32_t *__restrict__ node,
                                                 const int32_t *__restrict__ pos, const int32_t *__restrict__ count,
                                                 const int32_t *__restrict__ task_off,
                                                 const int32_t *__restrict__ child_first,
                                                 const uint8_t *__restrict__ node_err, const uint32_t *__restrict__ win,
                                                 SeqRef *__restrict__ order, Task *__restrict__ tasks,
                                                 LevelRec *__restrict__ cur, int32_t *__restrict__ status,
                                                 HintTables hints) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const int nd = RD(node, s);
    if (nd < lo || nd >= hi) return;
    const int c0 = RD(child_first, nd), k = RD(child_first, nd + 1) - c0, err = RD(node_err, nd);
    if (k <= 1 || err) {
        settle_trivial(c0, k, err, s, cur, status);
        return;
    }
    const int p = RD(pos, s);
    const int task = RD(task_off, nd) + p / TILE;
    SeqRef ref;
    ref.seq = (int32_t)s;
    ref.win = RD(win, s);
    WR(order, (size_t)task * TILE + (p & (TILE - 1)), ref);
    if ((p & (TILE - 1)) == 0) WR(tasks, task, make_task(nd, min(TILE, RD(count, nd) - p), c0, k, ref.win, hints));
}

// Level 0 needs no grouping: every sequence whose incoming status is OK stands on the root, so
// sequence s takes slot s (task s / 64).  Sequences that are not OK keep their slot with seq = -1
// (the walk treats such lanes as absent) and get their stop record here.
__global__ __launch_bounds__(256) void k_root_level(int64_t n, const int32_t *__restrict__ status_in,
                                                    int32_t *__restrict__ status, const uint32_t *__restrict__ win,
                                                    LevelRec *__restrict__ cur, const int32_t *__restrict__ child_first,
                                                    const uint8_t *__restrict__ node_err, SeqRef *__restrict__ order,
                                                    Task *__restrict__ tasks, int32_t *__restrict__ n_tasks,
                                                    HintTables hints) {
    const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const int st = status_in ? RD(status_in, s) : 0;
    WR(status, s, st);
    const int c0 = RD(child_first, 0), k = RD(child_first, 1) - c0, err = RD(node_err, 0);
    const bool trivial = k <= 1 || err;
    if (st != 0) WR(&cur[s].node, 0, -1);
    else if (trivial) settle_trivial(c0, k, err, s, cur, status);
    if (s == 0) WR(n_tasks, 0, trivial ? 0 : (int32_t)((n + TILE - 1) / TILE));
    if (trivial) return;
    SeqRef ref;
    ref.seq = st == 0 ? (int32_t)s : -1;
    ref.win = RD(win, s);
    WR(order, s, ref);
    if ((s & (TILE - 1)) == 0) WR(tasks, s / TILE, make_task(0, (int32_t)min((int64_t)TILE, n - s), c0, k, ref.win, hints));
}

// ------------------------------------------------------------------------------------------
// walk: one tree level.  stag/classify.py:22-52 for all sequences standing at nodes of level
// `depth`.  One workgroup = up to 64 sequences of ONE node (lane = sequence).
//
//  * Only the task's WINDOW of the rows is read: the union of the 64 rows' windows, rounded to
//    16-word chunks (64-byte sectors).  Outside its own window a row equals R0, so what a sibling
//    scores there is a constant of the sibling (gappre / gapsuf, added up on the host).
//    z = intercept + that constant + sum over the coefficients INSIDE the row's window whose
//    feature bit is set in the row.  For 253-column amplicon reads in a 1,500-column alignment
//    that is 256 B instead of 944 B per row and a sixth of the coefficients; for rows that differ
//    from R0 everywhere it degrades to the full row.
//  * The window slice of the rows is staged in LDS ([row][word], odd stride: lane l reading
//    word w is conflict-free); WCAP_CHUNKS chunks at a time, wider windows loop.
//  * One wave per sibling, NW siblings per pass.  A wave fetches its sibling's coefficients of
//    the slice with one 16-byte load per lane into its own piece of LDS and reads them back four
//    at a time by uniform-address ds_read_b128; per coefficient every lane adds w & -(bit):
//    v_add, ds_read_b32, v_bfe_i32, 2 x v_and, v_add_f64.
// ------------------------------------------------------------------------------------------
struct WalkArgs {
    const uint32_t *bits;       // [n][W]
    int W, NC;                  // NC = chunks per row
    const int32_t *n_tasks;
    const Task *tasks;
    const SeqRef *order;
    const double *intercept;
    const uint32_t *blk_off;    // [n_nodes*NC + 1]
    const double *gappre;       // [n_nodes][NC+1]: gap[0] + ... + gap[ch-1], added in that order (gap[ch] = what R0
                                // scores in chunk ch: the sum of the chunk's coefficients on "others" bits)
    const double *gapsuf;       // [n_nodes][NC+1]: gap[NC-1] + ... + gap[ch], added in that order
    const uint4 *blocks;        // Entry[4] per block
    LevelRec *lvl;              // [n] this level's records
    int32_t *status;
    const uint8_t *mfma_ok;     // [n_nodes] nodes whose tasks k_walk_mfma takes (null: none)
    unsigned long long *dbg;    // STAG_ABL == 10 builds: cycles per phase, summed over workgroups (else unused)
};
#if STAG_ABL == 10
#define STAG_STAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); \
                           phase_[i] += now_ - stamp_; stamp_ = now_; } while (0)
#else
#define STAG_STAMP(i) do { } while (0)
#endif

__device__ __forceinline__ double sigmoid(double z) { return 1.0 / (1.0 + exp(-z)); }

// 64-lane min / max with DPP moves (no LDS traffic); the result is wave-uniform.
__device__ __forceinline__ int wave_min_i(int v) {
    const int I = 0x7fffffff;
    v = min(v, __builtin_amdgcn_update_dpp(I, v, 0x111, 0xf, 0xf, false));  // row_shr:1
    v = min(v, __builtin_amdgcn_update_dpp(I, v, 0x112, 0xf, 0xf, false));  // row_shr:2
    v = min(v, __builtin_amdgcn_update_dpp(I, v, 0x114, 0xf, 0xf, false));  // row_shr:4
    v = min(v, __builtin_amdgcn_update_dpp(I, v, 0x118, 0xf, 0xf, false));  // row_shr:8
    v = min(v, __builtin_amdgcn_update_dpp(I, v, 0x142, 0xa, 0xf, false));  // row_bcast:15
    v = min(v, __builtin_amdgcn_update_dpp(I, v, 0x143, 0xc, 0xf, false));  // row_bcast:31
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_i(int v) {
    const int I = (int)0x80000000;
    v = max(v, __builtin_amdgcn_update_dpp(I, v, 0x111, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(I, v, 0x112, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(I, v, 0x114, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(I, v, 0x118, 0xf, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(I, v, 0x142, 0xa, 0xf, false));
    v = max(v, __builtin_amdgcn_update_dpp(I, v, 0x143, 0xc, 0xf, false));
    return __builtin_amdgcn_readlane(v, 63);
}

// (checked build) does the byte offset `off` of a row word lie in the chunks [sA, sB) that are staged?
__device__ __forceinline__ bool in_slice(uint32_t off, int sA, int sB) {
    return off >= (uint32_t)sA * (CHUNK_WORDS * 4) && off < (uint32_t)sB * (CHUNK_WORDS * 4) && (off & 3u) == 0u;
}

// entry en = {off, bit, w_lo, w_hi} in VGPRs (same value in every lane): w if this lane's row
// has the feature, else +0.0
__device__ __forceinline__ double masked_weight_v(const char *my_row, const uint4 en) {
    const uint32_t x = *reinterpret_cast<const uint32_t *>(my_row + en.x);
    const int m = __builtin_amdgcn_sbfe((int)x, en.y, 1u);  // 0 or -1
    return __hiloint2double((int)en.w & m, (int)en.z & m);
}

// LDS-DMA (global_load_lds_*): each active lane's 4 / 16 bytes at `src` land in LDS at dst + lane * 4 / 16
// (dst is wave-uniform) without passing through a VGPR.  hipcc counts these on vmcnt and drains them before
// a __syncthreads(); a wave that reads its OWN DMA'd data without a barrier waits with dma_wait().
#define STAG_GLOBAL __attribute__((address_space(1)))
#define STAG_LDS __attribute__((address_space(3)))
__device__ __forceinline__ void dma4(const uint32_t *src, uint32_t *dst) {
    __builtin_amdgcn_global_load_lds((const STAG_GLOBAL void *)src, (STAG_LDS void *)dst, 4, 0, 0);
}
__device__ __forceinline__ void dma16(const uint4 *src, uint4 *dst) {
    __builtin_amdgcn_global_load_lds((const STAG_GLOBAL void *)src, (STAG_LDS void *)dst, 16, 0, 0);
}
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_reads_done() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// LDS of one walk workgroup: 20 KB (8 workgroups per CU), cut per task into the staged rows -- 64 rows of
// `stride` words, stride = the staged width + 1 (odd: lane l reading word w is conflict-free) -- and, behind them,
// NW equal wave-private pieces for coefficients, each used as two halves: the sibling a wave scores reads one
// half while the coefficients of its next sibling land in the other.
constexpr int WALK_LDS_WORDS = 5120;
constexpr int RESULT_SLOT_ENTRIES = TILE * 12 / 16;   // a wave's result slot (64 doubles + 64 ints) in entries
static_assert((WALK_LDS_WORDS - TILE * ROW_STRIDE) / NW / 4 >= RESULT_SLOT_ENTRIES, "result slot must fit a wave's piece");

__global__ __launch_bounds__(NW * 64, 8) void k_walk_level(WalkArgs a) {
    __shared__ __attribute__((aligned(16))) uint32_t s_mem[WALK_LDS_WORDS];

    const int lane = threadIdx.x & 63;
    // The grid has a workgroup for every task there COULD be (the host does not know n_tasks): the surplus ones
    // leave after one scalar load.  (Fetching the task record and the sequences before this test, so that all three
    // round trips overlap, was measured: the surplus workgroups -- a third of the grid on the deep levels -- then
    // each start two fetches they never use, and the C2 walk takes 4 % longer.)
    const int n_tasks = RDC(a.n_tasks, 0);
    if ((int)blockIdx.x >= n_tasks) return;
    // the task's slots of order[] sit at a fixed place: fetched together with the task record, not after it
    // (slots past the task's count hold stale but allocated data)
    const i32x8 tk = RDC(reinterpret_cast<const i32x8 *>(a.tasks), 2 * blockIdx.x);
    SeqRef me = RD(a.order, (size_t)blockIdx.x * TILE + lane);
    asm volatile("" : : "s"(tk[0]), "v"(me.seq), "v"(me.win));   // (left alone, the compiler starts the second after the first has arrived)
    const uint32_t STAG_CONST *blk_off = as_const(a.blk_off);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c0 = tk[2], k = tk[3];
    if (k <= 1) return;   // never emitted: k_scatter / k_root_level settle such nodes themselves
    if (tk[6]) return;   // k_walk_mfma's task
    if (lane >= tk[1]) {
        me.seq = -1;
        me.win = 0xffffu;
    }
    const int seq = me.seq;
    auto put = [&](int node_out, double prob) {
        LevelRec r;
        r.node = node_out;
        r.pad = 0;
        r.prob = prob;
        WR(a.lvl, seq, r);
    };
#if STAG_ABL == 5
    if (seq >= 0) put(c0, 0.5);
    return;
#endif

    // ---- the task's window in chunks [cA, cB): union of its rows' windows
    int lo = 0x7fff, hi = 0;
    int own_lo = 0, own_hi = 0;   // this lane's own window in chunks (empty: its row is R0 everywhere)
    if (seq >= 0) {
        const uint32_t wn = me.win;
        const int l4 = (int)(wn & 0xffffu), h4 = (int)(wn >> 16);
        if (h4 > l4) {   // 4-word units -> 16-word chunks
            lo = own_lo = l4 >> 2;
            hi = own_hi = min((h4 + 3) >> 2, a.NC);
        }
    }
    const int cA = wave_min_i(lo);
    const int cB = min(wave_max_i(hi), a.NC);
    const int wA = cB > cA ? cA : 0, wB = cB > cA ? cB : 0;   // empty window: every chunk is "outside"
    const bool one_slice = wB - wA <= WCAP_CHUNKS;            // the usual case: staged once for all siblings

    // ---- LDS layout of this task
    // (wider windows go slice by slice, WIDE_SLICE_CHUNKS at a time: narrower slices than the buffer could hold, so
    // that each wave's piece holds the slice's coefficients of TWO siblings, fetched together with the rows)
    constexpr int WIDE_SLICE_CHUNKS = 3;
    const int stride = one_slice ? (((wB - wA) * CHUNK_WORDS) | 1) : (WIDE_SLICE_CHUNKS * CHUNK_WORDS) | 1;
    const int ent_base_words = (TILE * stride + 3) & ~3;
    const int piece = (WALK_LDS_WORDS - ent_base_words) / (4 * NW);      // entries per wave
    const int half = min(piece >> 1, 64) & ~3;                          // entries per buffer: one DMA (one per lane) fills it
    uint4 *const s_ent = reinterpret_cast<uint4 *>(s_mem + ent_base_words) + wave * piece;
    uint32_t *const s_rows = s_mem;

    // CANONICAL SUMMATION.  A row's OWN window (from its own bits: the chunks in which it differs
    // from R0) fixes how its score is added up: z = (gappre[own_lo] + gapsuf[own_hi]) -- what R0
    // scores before and after that window, each a chain the host added up in a fixed order -- plus,
    // in increasing chunk order, one term per chunk of the own window: the lane's sum over that
    // chunk's coefficients (in stored order, one accumulator).  Chunks that are inside the task's
    // window only because of other rows are left out (they are part of the two constants).  So a
    // score depends only on (row, node): not on which other rows share the task.  That keeps results
    // bit-identical from run to run (task composition comes from atomics) and makes identical
    // sibling rows tie exactly, so the first stored child wins as in classify.py:41.
    double best_p = -1.0;
    int best_c = -1;
    // rows of the slice [sA, sB) -> LDS by LDS-DMA: one row per wave and step, lanes over words, no VGPR in
    // between.  The caller makes sure nobody still reads the previous slice (barrier) before, and drains the
    // DMA + barriers after.  Rows of absent lanes (seq < 0) and words past the slice keep stale data that
    // no entry of the slice refers to / whose result is never stored.
    // (Each lane works out the address of ITS row once; a row's address then reaches the wave that stages it as two
    // scalars: per row two v_readlane, the LDS address and the DMA -- a quarter of the instructions that computing
    // the 64-bit address from the row's sequence id in scalar code takes.  An absent lane stands for row 0: staged
    // like any other, never scored.)
    const uint32_t *const my_row_words = a.bits + (size_t)max(seq, 0) * a.W;
    auto stage_rows = [&](int sA, int sB) {
        const int w0 = sA * CHUNK_WORDS;
        const int width = min(sB * CHUNK_WORDS, a.W) - w0;   // <= WCAP_CHUNKS * 16 words
        const uint64_t at = reinterpret_cast<uint64_t>(my_row_words + w0);
        int at_lo = (int)(uint32_t)at, at_hi = (int)(uint32_t)(at >> 32);
        // EVERY lane needs its address in registers before the lanes past the slice's width are switched off: the
        // readlanes below read lanes that are inactive by then, and without this the compiler moves the address
        // arithmetic into the masked region (their registers would then hold whatever was there before)
        asm volatile("" : "+v"(at_lo), "+v"(at_hi));
        if (lane < width) {
            uint32_t *dst = &s_rows[wave * stride];
#pragma unroll
            for (int rr = 0; rr < TILE / NW; ++rr) {
                const uint32_t r_lo = (uint32_t)__builtin_amdgcn_readlane(at_lo, rr * NW + wave);
                const uint32_t r_hi = (uint32_t)__builtin_amdgcn_readlane(at_hi, rr * NW + wave);
                const uint32_t *src = reinterpret_cast<const uint32_t *>(((uint64_t)r_hi << 32) | r_lo) + lane;
                if (GUARD(src, 4)) dma4(src, dst);
                dst += NW * stride;
            }
        }
    };
    // up to `half` coefficients from e_lo on -> buffer `buf` of this wave's piece (wave-private)
    auto fetch_ent = [&](uint32_t e_lo, uint32_t e_hi, int buf) {
        if (lane < half && e_lo + lane < e_hi && GUARD(a.blocks + e_lo + lane, 16)) dma16(a.blocks + e_lo + lane, s_ent + buf * half);
    };
    // z += the sibling's chunk sums over the slice [sA, sB); its first `half` coefficients are in buffer `buf`
    auto score_slice = [&](double &z, const uint32_t STAG_CONST *bo, uint32_t e_lo, uint32_t e_hi, int sA, int sB, int buf) {
        // which chunks of this slice lie in the lane's own window (bit i: chunk sA + i)
        const uint32_t own_mask = (own_hi > sA ? (own_hi - sA >= 32 ? 0xffffffffu : (1u << (own_hi - sA)) - 1u) : 0u) &
                                  ~(own_lo > sA ? (own_lo - sA >= 32 ? 0xffffffffu : (1u << (own_lo - sA)) - 1u) : 0u);
        // entry offsets are relative to the row; the slice starts at chunk sA
        const char *my_row = reinterpret_cast<const char *>(s_rows) + (lane * stride * 4 - sA * (CHUNK_WORDS * 4));
        const uint4 *ent = s_ent + buf * half;
        if (e_hi - e_lo <= (uint32_t)half) {
            // the usual case: every coefficient of the slice is in the buffer
            for (int ch = sA; ch < sB; ++ch) {
                const uint32_t e0 = RDK(bo, ch) * BLOCK_ENTRIES - e_lo, e1 = RDK(bo, ch + 1) * BLOCK_ENTRIES - e_lo;
                if (!CHECK_THAT(e0 <= e1 && e1 <= (uint32_t)half, e0, e1)) break;
                double cs = 0.0;
                for (uint32_t e = e0; e < e1; e += BLOCK_ENTRIES) {
                    const uint4 *pe = &ent[e];   // wave-uniform address: LDS broadcast
                    const uint4 n0 = pe[0], n1 = pe[1], n2 = pe[2], n3 = pe[3];
                    if (!CHECK_THAT(in_slice(n0.x, sA, sB) && in_slice(n1.x, sA, sB) && in_slice(n2.x, sA, sB) && in_slice(n3.x, sA, sB), n0.x, sA)) break;
                    cs += masked_weight_v(my_row, n0);
                    cs += masked_weight_v(my_row, n1);
                    cs += masked_weight_v(my_row, n2);
                    cs += masked_weight_v(my_row, n3);
                }
                z += ((own_mask >> (ch - sA)) & 1u) ? cs : 0.0;   // z is never -0.0: adding +0.0 is exact
            }
        } else {
            uint32_t first = e_lo;   // first entry held in the buffer
            for (int ch = sA; ch < sB; ++ch) {
                const uint32_t e0 = RDK(bo, ch) * BLOCK_ENTRIES, e1 = RDK(bo, ch + 1) * BLOCK_ENTRIES;
                if (!CHECK_THAT(e_lo <= e0 && e0 <= e1 && e1 <= e_hi, e0, e1)) break;
                double cs = 0.0;
                for (uint32_t e = e0; e < e1; e += BLOCK_ENTRIES) {
                    if (e + BLOCK_ENTRIES > first + (uint32_t)half) {   // past the buffer: next stretch
                        first = e;
                        lds_reads_done();
                        fetch_ent(first, e_hi, buf);
                        dma_wait();
                    }
                    const uint4 *pe = &ent[e - first];
                    const uint4 n0 = pe[0], n1 = pe[1], n2 = pe[2], n3 = pe[3];
                    if (!CHECK_THAT(e >= first && in_slice(n0.x, sA, sB) && in_slice(n1.x, sA, sB) && in_slice(n2.x, sA, sB) && in_slice(n3.x, sA, sB), n0.x, sA)) break;
                    cs += masked_weight_v(my_row, n0);
                    cs += masked_weight_v(my_row, n1);
                    cs += masked_weight_v(my_row, n2);
                    cs += masked_weight_v(my_row, n3);
                }
                z += ((own_mask >> (ch - sA)) & 1u) ? cs : 0.0;
            }
        }
    };
    auto gap_consts = [&](int c) {   // two 8-byte gathers per lane; lanes of one task mostly share the addresses
        const size_t base = (size_t)c * (a.NC + 1);
        int i_lo = own_lo, i_hi = own_hi;
        asm volatile("" : "+v"(i_lo), "+v"(i_hi));   // keeps the two 64-bit addresses out of long-lived registers
        return RD(a.gappre, base + i_lo) + RD(a.gapsuf, base + i_hi);
    };
    auto decide = [&](double z, int c) {
#if STAG_ABL == 2
        const double p = z + RDC(a.intercept, c);
#else
        const double p = sigmoid(z + RDC(a.intercept, c));
#endif
        if (p > best_p) { best_p = p; best_c = c; }  // strict '>' in stored order
    };

    if (one_slice) {
        // ---- the usual case: the window fits the staging buffer and is staged once for all siblings.  A wave
        // scores the siblings wave, wave + NW, ...; while it scores one, the coefficients of its next one land in
        // the other half of its piece, and the chunk offsets of the one after that are on their way.
        const int n_mine = k > wave ? (k - wave + NW - 1) / NW : 0;
        const bool windowed = wB > wA;
        uint32_t e_lo = 0, e_hi = 0, ne_lo = 0, ne_hi = 0;     // this sibling's / the next sibling's coefficient range
        double z = 0.0, nz = 0.0;
        if (n_mine > 0) {
#if STAG_ABL != 4
            z = gap_consts(c0 + wave);
#endif
            if (windowed) {
                const uint32_t STAG_CONST *bo = blk_off + (size_t)(c0 + wave) * a.NC;
                e_lo = RDK(bo, wA) * BLOCK_ENTRIES;
                e_hi = RDK(bo, wB) * BLOCK_ENTRIES;
            }
        }
        if (windowed) {
#if STAG_ABL != 3
            stage_rows(wA, wB);
#endif
            fetch_ent(e_lo, e_hi, 0);        // in flight together with the rows
            if (n_mine > 1) {
                const uint32_t STAG_CONST *bo = blk_off + (size_t)(c0 + wave + NW) * a.NC;
                ne_lo = RDK(bo, wA) * BLOCK_ENTRIES;
                ne_hi = RDK(bo, wB) * BLOCK_ENTRIES;
            }
            __syncthreads();                 // drains the DMA of every wave, then the barrier
        }
        for (int i = 0; i < n_mine; ++i) {
            const int c = c0 + wave + i * NW;
            const uint32_t STAG_CONST *bo = blk_off + (size_t)c * a.NC;
            uint32_t nn_lo = 0, nn_hi = 0;
            if (i + 1 < n_mine) {
#if STAG_ABL != 4
                nz = gap_consts(c + NW);
#endif
                if (windowed) {
                    fetch_ent(ne_lo, ne_hi, (i + 1) & 1);        // that half's last reader was sibling i - 1
                    if (i + 2 < n_mine) {
                        const uint32_t STAG_CONST *bo2 = blk_off + (size_t)(c + 2 * NW) * a.NC;
                        nn_lo = RDK(bo2, wA) * BLOCK_ENTRIES;
                        nn_hi = RDK(bo2, wB) * BLOCK_ENTRIES;
                    }
                }
            }
#if STAG_ABL != 1
            if (windowed) score_slice(z, bo, e_lo, e_hi, wA, wB, i & 1);
#endif
            decide(z, c);
            if (i + 1 < n_mine) {
                lds_reads_done();
                dma_wait();                  // the next sibling's coefficients have landed
                z = nz; e_lo = ne_lo; e_hi = ne_hi; ne_lo = nn_lo; ne_hi = nn_hi;
            }
        }
    } else {
        // ---- a window wider than the staging buffer: slice after slice; each slice is staged once for
        // two passes (8 siblings), whose sums wait in registers
        constexpr int PG = 2;   // 4 measured slower: the sums then spill
        for (int pg0 = 0; pg0 < k; pg0 += PG * NW) {
            double zz[PG];
#pragma unroll
            for (int j = 0; j < PG; ++j) {
                zz[j] = 0.0;
                if (pg0 + j * NW + wave < k) zz[j] = gap_consts(c0 + pg0 + j * NW + wave);
            }
            for (int sA = wA; sA < wB; sA += WIDE_SLICE_CHUNKS) {
                const int sB = min(wB, sA + WIDE_SLICE_CHUNKS);
                uint32_t e_lo[PG], e_hi[PG];
#pragma unroll
                for (int j = 0; j < PG; ++j) {
                    e_lo[j] = e_hi[j] = 0;
                    if (pg0 + j * NW + wave < k) {   // wave-uniform
                        const uint32_t STAG_CONST *bo = blk_off + (size_t)(c0 + pg0 + j * NW + wave) * a.NC;
                        e_lo[j] = RDK(bo, sA) * BLOCK_ENTRIES;
                        e_hi[j] = RDK(bo, sB) * BLOCK_ENTRIES;
                    }
                }
                __syncthreads();   // everyone is done reading the previous slice (rows and coefficients)
                stage_rows(sA, sB);
#pragma unroll
                for (int j = 0; j < PG; ++j) fetch_ent(e_lo[j], e_hi[j], j);   // in flight together with the rows
                __syncthreads();   // DMA drained, rows visible
#pragma unroll
                for (int j = 0; j < PG; ++j) {
                    if (pg0 + j * NW + wave < k) {
                        const uint32_t STAG_CONST *bo = blk_off + (size_t)(c0 + pg0 + j * NW + wave) * a.NC;
                        score_slice(zz[j], bo, e_lo[j], e_hi[j], sA, sB, j);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < PG; ++j)
                if (pg0 + j * NW + wave < k) decide(zz[j], c0 + pg0 + j * NW + wave);
        }
    }
    // the result slots reuse each wave's own coefficient piece: LDS operations of one wave stay in order, and no
    // DMA is in flight any more
    double *const my_z = reinterpret_cast<double *>(s_ent);
    int *const my_c = reinterpret_cast<int *>(s_ent) + 2 * TILE;
    my_z[lane] = best_p;
    my_c[lane] = best_c;
    __syncthreads();
    if (wave == 0 && seq >= 0) {
        for (int w = 1; w < NW; ++w) {  // first maximum in stored order = largest p, then smallest id
            const double p = reinterpret_cast<const double *>(s_ent + w * piece)[lane];
            const int cc = (reinterpret_cast<const int *>(s_ent + w * piece) + 2 * TILE)[lane];
            if (cc >= 0 && (p > best_p || (p == best_p && cc < best_c) || best_c < 0)) { best_p = p; best_c = cc; }
        }
        // best_c < 0: every probability was NaN, the reference stops here (classify.py:49)
        put(best_c, best_c >= 0 ? best_p : 0.0);
    }
}

// ------------------------------------------------------------------------------------------
// walk on the matrix cores: the same level for tasks of nodes that have the tables of stag_db_create's
// "matrix-core tables".  For the task's 64 rows and its window, X[row][u] = bit of union feature u (one byte,
// 0 or 1; 0 outside the row's OWN window) is built in LDS from the staged row words -- one bit extraction per
// (row, union feature), shared by all k children, where the gather walk does one per (row, child, coefficient)
// -- and  C[row][6 j + s] = sum_u X[row][u] * digit_s(w[j][u])  is computed by v_mfma_i32_16x16x64_i8.
// z[row][j] = intercept + gappre[own_lo] + gapsuf[own_hi] + 2^-q * sum_s 256^s C[row][6 j + s]: integer
// arithmetic up to the last step, so a score depends only on (row, node) -- never on the other rows of the
// task -- and identical sibling rows tie exactly.  The coefficients are rounded to multiples of 2^-q (q >= 39
// for |w| < 128): z is within n_coefficients * 2^-(q+1) of the exact sum, far inside the 1e-9 the path allows.
// The child is chosen by comparing scores where that is the same as comparing probabilities, and probabilities
// where it is not (beats()); only then, and for the winner, a sigmoid is computed.
// ------------------------------------------------------------------------------------------
struct MfmaArgs {
    const uint32_t *pf_off;     // [n_nodes * NC + 1] first union feature of (parent, chunk), multiples of 16
    const uint32_t *pfeat;      // descriptor per union feature: byte offset of the word in the row | bit << 16
    const int8_t *dig;          // digit matrices
    const uint64_t *dig_base;   // [n_nodes] where the parent's matrix starts
    const uint32_t *dig_ld;     // [n_nodes] features of the parent = bytes per column
    const double *qscale;       // [n_nodes] 2^-q
    const uint8_t *mfma_ok;     // [n_nodes] 1: this kernel walks the node's tasks, 0: k_walk_level does
};
using v4i = __attribute__((ext_vector_type(4))) int;

constexpr int MW_LDS_WORDS = 5632;                // 22,528 B
constexpr int MW_C_STRIDE = 65;                   // C tile of a round: [64 rows][4 column blocks = 8 children] + 1
// Is sigmoid(z) > sigmoid(zb) as fp64 numbers (what classify.py:41 compares)?  The sigmoid is increasing, so the
// scores themselves decide -- except where different scores round to the same probability: towards 1.0 (two scores
// above 20 can share a probability; from 36.74 on it is exactly 1.0) and where exp(-z) overflows (probability 0.0
// below -709.78).  There the two probabilities are computed and compared, so that such siblings tie and the first
// stored one wins, as in the reference.  (Between those bounds two scores collide only when they differ by less
// than 5e-8, i.e. when the probabilities agree to 16 digits.)
constexpr double Z_EXACT_ABOVE = 20.0, Z_EXACT_BELOW = -700.0;
__device__ __forceinline__ bool beats(double z, double zb) {
    const double top = fmax(z, zb);
    if (top < Z_EXACT_ABOVE && top > Z_EXACT_BELOW) return z > zb;
    return sigmoid(z) > sigmoid(zb);
}
static_assert(TILE * MW_C_STRIDE <= MW_LDS_WORDS, "the C tile overlays rows + X");
static_assert(MFMA_MAX_K <= 16, "two rounds of 8 children, one column block per wave and round");

// Workgroup = one task of <= 64 rows, 4 waves.
//   1. every fetch the task needs is issued as soon as its record is there (the record carries the hint, see Task):
//      row words by LDS-DMA, this wave's feature descriptors (scalar), its digit fragments and the constants of the
//      children it will score;
//   2. X: lane = row, each wave a quarter of the features (descriptors in scalar registers), 16 bytes per store;
//   3. C = X * digits: wave w multiplies all 64 rows with column blocks w (children 2w, 2w+1) and w + 4;
//   4. through a C tile in LDS (it overlays rows and X) to lane = (row, group): wave w scores ITS 16 rows, lane group
//      g the children g, g + 4, ...; the four groups are reduced with shuffles; the wave writes its rows' records.
__global__ __launch_bounds__(NW * 64, 5) void k_walk_mfma(WalkArgs a, MfmaArgs m) {
    __shared__ __attribute__((aligned(16))) uint32_t s_mem[MW_LDS_WORDS];

    const int lane = threadIdx.x & 63;
#if STAG_ABL == 10
    unsigned long long phase_[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_ = __builtin_readcyclecounter();
#endif
    const int n_tasks = RDC(a.n_tasks, 0);
    const i32x8 t0 = RDC(reinterpret_cast<const i32x8 *>(a.tasks), 2 * blockIdx.x);
    const i32x8 t2 = RDC(reinterpret_cast<const i32x8 *>(a.tasks), 2 * blockIdx.x + 1);
    SeqRef me = RD(a.order, (size_t)blockIdx.x * TILE + lane);     // lane = row of the task (staging, X)
    if ((int)blockIdx.x >= n_tasks) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int node = t0[0], count = t0[1], c0 = t0[2], k = t0[3];
    if (k <= 1 || !t0[6]) return;                              // the gather kernel's task
    if (lane >= count) {
        me.seq = -1;
        me.win = 0xffffu;
    }
    const int seq = me.seq;

    // ---- windows (as in k_walk_level)
    int lo = 0x7fff, hi = 0, own_lo = 0, own_hi = 0;
    if (seq >= 0) {
        const uint32_t wn = me.win;
        const int l4 = (int)(wn & 0xffffu), h4 = (int)(wn >> 16);
        if (h4 > l4) {
            lo = own_lo = l4 >> 2;
            hi = own_hi = min((h4 + 3) >> 2, a.NC);
        }
    }
    const int cA = wave_min_i(lo);
    const int cB = min(wave_max_i(hi), a.NC);
    const int wA = cB > cA ? cA : 0, wB = cB > cA ? cB : 0;
    // do all rows of the task have the task's window?  (amplicon reads: yes)  Then no bit has to be masked.
    const bool uniform_win = __ballot(seq >= 0 && (own_lo != wA || own_hi != wB)) == 0ull;
#if STAG_ABL == 6
    if (wave == 0 && seq >= 0) { LevelRec rec; rec.node = c0; rec.pad = wA + wB; rec.prob = 0.5; a.lvl[seq] = rec; }
    return;
#endif
    STAG_STAMP(0);   // task record + order arrived, windows known
    // does the hint describe this window?  Otherwise (rows with other windows joined the task, or the window is
    // staged slice by slice) the lookups happen here, one after the other.
    const bool hinted = wA == t0[4] && wB == t0[5] && wB - wA <= WCAP_CHUNKS;
    uint32_t U = (uint32_t)t2[2];
    double qs = __hiloint2double(t2[7], t2[6]);
    const int8_t *dig_node = nullptr;                           // column 0, first feature of the node
    uint32_t pf_first = 0;
    if (!hinted) {
        U = RDC(m.dig_ld, node);
        qs = RDC(m.qscale, node);
        dig_node = m.dig + RDC(m.dig_base, node);
        pf_first = RDC(m.pf_off, (size_t)node * a.NC);
    }

    // ---- LDS: staged row words | X bytes; the C tile later overlays both
    const int slice_chunks = min(wB - wA, WCAP_CHUNKS);
    const int stride = (slice_chunks * CHUNK_WORDS) | 1;
    const int rows_words = (TILE * stride + 3) & ~3;
    const int XK = (MW_LDS_WORDS - rows_words) * 4 >= TILE * (128 + 16) ? 128 : 64;   // union features per X build
    const int XS = XK + 16;                                                             // bytes per X row
    uint32_t *const s_rows = s_mem;
    uint8_t *const s_x = reinterpret_cast<uint8_t *>(s_mem + rows_words);
    const int per_wave = XK / NW;                              // 32 or 16 features per wave and X build

    // ---- the children this lane scores at the end: row r of this wave's 16, group g -> children g, g + 4, g + 8, g + 12
    const int r = lane & 15, g = lane >> 4;
    const int my_slot = 16 * wave + r;
    const int myseq = __shfl(seq, my_slot);
    const int r_lo = __shfl(own_lo, my_slot), r_hi = __shfl(own_hi, my_slot);
    // (what R0 scores outside the row's window) + intercept for this lane's children of a round (g and g + 4 of the
    // round's 8).  Branch-free -- children past the last one re-read the last one's constants and are never scored --
    // so that the six loads are issued back to back and waited for once.
    auto child_consts = [&](int rd, double (&kc)[2]) {
        double gp[2], gs[2], ic[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = c0 + min(8 * rd + 4 * h + g, k - 1);
            const size_t base = (size_t)c * (a.NC + 1);
            gp[h] = RD(a.gappre, base + r_lo);
            gs[h] = RD(a.gapsuf, base + r_hi);
            ic[h] = RD(a.intercept, c);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) kc[h] = (gp[h] + gs[h]) + ic[h];
    };
    double kc0[2];
    child_consts(0, kc0);

    const int ncb = (k + 1) >> 1;                              // column blocks of 16 (two children each)
    int *const ctile = reinterpret_cast<int *>(s_mem);
    double best_raw = 0.0;
    int best_c = -1;
    // A round = 8 children = column blocks 4 rd .. 4 rd + 3, one per wave.  Nodes with more than 8 children take a
    // second round that stages and builds X again (the C tile of the first round has overwritten them).
    for (int rd = 0; 8 * rd < k; ++rd) {
        const int cb = wave + rd * NW;                         // this wave's column block of the round
        double kc[2] = {kc0[0], kc0[1]};
        if (rd) child_consts(1, kc);
        v4i acc[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = v4i{0, 0, 0, 0};
        for (int sA = wA; sA < wB; sA += WCAP_CHUNKS) {
            const int sB = min(wB, sA + WCAP_CHUNKS);
            if (sA != wA || rd) __syncthreads();           // the previous slice's X / rows (or the C tile) have been read
            {   // row words of the slice -> LDS (LDS-DMA; lanes over words, one row per wave and step)
                // (per-lane row addresses handed to the staging wave by v_readlane, as in k_walk_level: an absent
                // lane stands for row 0, and every lane has its address before the lanes past the width go off)
                const int w0 = sA * CHUNK_WORDS;
                const int width = min(sB * CHUNK_WORDS, a.W) - w0;
                const uint64_t at = reinterpret_cast<uint64_t>(a.bits + (size_t)max(seq, 0) * a.W + w0);
                int at_lo = (int)(uint32_t)at, at_hi = (int)(uint32_t)(at >> 32);
                asm volatile("" : "+v"(at_lo), "+v"(at_hi));
                if (lane < width) {
                    uint32_t *dst = &s_rows[wave * stride];
#pragma unroll
                    for (int rr = 0; rr < TILE / NW; ++rr) {
                        const uint32_t a_lo = (uint32_t)__builtin_amdgcn_readlane(at_lo, rr * NW + wave);
                        const uint32_t a_hi = (uint32_t)__builtin_amdgcn_readlane(at_hi, rr * NW + wave);
#if STAG_ABL != 7
                        const uint32_t *src = reinterpret_cast<const uint32_t *>(((uint64_t)a_hi << 32) | a_lo) + lane;
                        if (GUARD(src, 4)) dma4(src, dst);
#endif
                        dst += NW * stride;
                    }
                }
            }
            uint32_t f0, f1;
            const int8_t *dig_win;                         // column 0, first feature of the slice
            if (hinted) {
                f0 = (uint32_t)t2[3];
                f1 = f0 + (uint32_t)t2[4];
                dig_win = m.dig + (((uint64_t)(uint32_t)t2[1] << 32) | (uint32_t)t2[0]);
            } else {
                f0 = RDC(m.pf_off, (size_t)node * a.NC + sA);
                f1 = RDC(m.pf_off, (size_t)node * a.NC + sB);
                dig_win = dig_node + (f0 - pf_first);
            }
            // this lane's own chunks inside the slice (bit i: chunk sA + i)
            const uint32_t own_mask = (own_hi > sA ? (own_hi - sA >= 32 ? 0xffffffffu : (1u << (own_hi - sA)) - 1u) : 0u) &
                                      ~(own_lo > sA ? (own_lo - sA >= 32 ? 0xffffffffu : (1u << (own_lo - sA)) - 1u) : 0u);
            const char *my_row = reinterpret_cast<const char *>(s_rows) + (lane * stride * 4 - sA * (CHUNK_WORDS * 4));
            if (!CHECK_THAT(f0 <= f1 && f1 - f0 <= U, f0, f1)) break;
            for (uint32_t x0 = f0; x0 < f1; x0 += XK) {
                // descriptors of this wave's features (scalar) and its digit fragments: in flight across the barrier
                const uint32_t u0 = x0 + wave * per_wave;
                i32x16 d0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, d1 = d0;
                if (u0 < f1) d0 = RDC(reinterpret_cast<const i32x16 *>(m.pfeat + u0), 0);
                if (per_wave > 16 && u0 + 16 < f1) d1 = RDC(reinterpret_cast<const i32x16 *>(m.pfeat + u0 + 16), 0);
                const int nkb = min((int)((f1 - x0 + 63) >> 6), XK >> 6);
                // branch-free (a wave without a column block, or a second 64-feature block that does not exist, re-reads
                // a fragment that exists; those MFMAs are skipped below): loads in a branch would be waited for at its end
                v4i bf[2];
                {
                    const int8_t *bp = dig_win + (size_t)(16 * min(cb, ncb - 1) + (lane & 15)) * U + (x0 - f0) + 16 * (lane >> 4);
                    bf[0] = RD(reinterpret_cast<const v4i *>(bp), 0);
                    bf[1] = RD(reinterpret_cast<const v4i *>(bp + (nkb > 1 ? 64 : 0)), 0);
                }
                STAG_STAMP(1);   // everything issued
                __syncthreads();                           // rows landed (first build) / the previous X has been multiplied
                STAG_STAMP(2);   // rows, descriptors, fragments arrived
                // ---- X: lane = row; 16 features -> 16 bytes -> one 16-byte LDS store
                auto build16 = [&](const i32x16 &d, bool valid, int at) {
                    uint32_t pk[4] = {0u, 0u, 0u, 0u};
                    if (valid) {                           // (f1 is a multiple of 16: a group is in or out)
#pragma unroll
                        for (int h8 = 0; h8 < 16; h8 += 8) {   // eight LDS reads in flight together
                            uint32_t word[8];
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                if (!CHECK_THAT(in_slice((uint32_t)d[h8 + i] & 0xffffu, sA, sB), d[h8 + i], sA)) { word[i] = 0u; continue; }
                                word[i] = *reinterpret_cast<const uint32_t *>(my_row + ((uint32_t)d[h8 + i] & 0xffffu));
                            }
                            if (uniform_win) {
#pragma unroll
                                for (int i = 0; i < 8; ++i)
                                    pk[(h8 + i) >> 2] |= ((word[i] >> ((uint32_t)d[h8 + i] >> 16)) & 1u) << (8 * (i & 3));
                            } else {
#pragma unroll
                                for (int i = 0; i < 8; ++i) {
                                    const uint32_t desc = (uint32_t)d[h8 + i];
                                    const uint32_t bit = (word[i] >> (desc >> 16)) & (own_mask >> (((desc & 0xffffu) >> 6) - sA)) & 1u;
                                    pk[(h8 + i) >> 2] |= bit << (8 * (i & 3));
                                }
                            }
                        }
                    }
                    *reinterpret_cast<uint4 *>(s_x + lane * XS + at) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                };
#if STAG_ABL != 8
                build16(d0, u0 < f1, wave * per_wave);
                if (per_wave > 16) build16(d1, u0 + 16 < f1, wave * per_wave + 16);
#endif
                STAG_STAMP(3);   // X built
                __syncthreads();                           // X complete
                STAG_STAMP(4);
                // ---- C += X * digits: all 64 rows, this wave's column block
                if (cb < ncb) {
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb)
                        if (kb < nkb) {
#pragma unroll
                            for (int rb = 0; rb < 4; ++rb) {
                                const v4i af = *reinterpret_cast<const v4i *>(s_x + (16 * rb + (lane & 15)) * XS + 64 * kb + 16 * (lane >> 4));
                                acc[rb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af, bf[kb], acc[rb], 0, 0, 0);
                            }
                        }
                }
            }
        }
        STAG_STAMP(5);   // MFMAs issued
        // ---- scores of the round's 8 children through the C tile: [64 rows][4 column blocks]
        __syncthreads();                                   // X has been read by everyone
        if (cb < ncb) {
            // C/D layout of a 16x16 tile: column = lane & 15, rows 4 (lane >> 4) .. + 3 in the 4 registers
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    ctile[(16 * rb + 4 * (lane >> 4) + q) * MW_C_STRIDE + 16 * wave + (lane & 15)] = acc[rb][q];
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {                      // this lane's children 8 rd + g and 8 rd + 4 + g
            const int j = 8 * rd + 4 * h + g;
            if (j < k) {
                const int *cr = ctile + my_slot * MW_C_STRIDE + 8 * (4 * h + g);
                double v = (double)cr[5];
#pragma unroll
                for (int sl = 4; sl >= 0; --sl) v = v * 256.0 + (double)cr[sl];
                const double z = kc[h] + v * qs;
                if (best_c < 0 || beats(z, best_raw)) { best_raw = z; best_c = c0 + j; }   // strict '>' in stored order
            }
        }
    }
    STAG_STAMP(6);   // tile turned around, scores
    // ---- first maximum in stored order over the four lane groups of a row: largest probability, then smallest child
#pragma unroll
    for (int step = 16; step <= 32; step <<= 1) {
        const double oraw = __shfl_xor(best_raw, step);
        const int oc = __shfl_xor(best_c, step);
        if (oc >= 0 && (best_c < 0 || beats(oraw, best_raw) || (oc < best_c && !beats(best_raw, oraw)))) { best_raw = oraw; best_c = oc; }
    }
    if (g == 0 && myseq >= 0) {
        LevelRec rec;
        rec.node = best_c;
        rec.pad = 0;
        rec.prob = sigmoid(best_raw);
        WR(a.lvl, myseq, rec);
    }
    STAG_STAMP(7);   // reduction, sigmoid, store
#if STAG_ABL == 10
    if (threadIdx.x == 0) {      // one slot of 16 counters per 4096 workgroups: no hot address
        unsigned long long *slot = a.dbg + 16 * (blockIdx.x & 4095);
        for (int i = 0; i < 8; ++i) atomicAdd(slot + i, phase_[i]);
        atomicAdd(slot + 15, 1ull);
    }
#endif
}

// ------------------------------------------------------------------------------------------
// rank cut: classify.py:106-110 (placeholder fix-up) and :58-72 (find_correct_level)
// ------------------------------------------------------------------------------------------
// One workgroup = 256 sequences, one thread per sequence.  A thread gathers its records of all
// levels (lvl[d][s]: coalesced across the workgroup), works on its own slice of LDS, and the
// [256][D] / [256][K] tiles then leave through coalesced stores (a thread storing its own 56-byte
// slice of HBM would cost one cache line per lane).
__global__ __launch_bounds__(256) void k_rankcut(int64_t n, int D, int levels, int K, int n_tf,
                                                 const double *__restrict__ tf, const LevelRec *__restrict__ lvl,
                                                 int32_t *__restrict__ path_node, double *__restrict__ path_prob,
                                                 int32_t *__restrict__ path_len, int32_t *__restrict__ n_lead_single,
                                                 double *__restrict__ level_prob, int32_t *__restrict__ selected,
                                                 int32_t *__restrict__ status) {
    extern __shared__ double s_rc[];   // [256][D] path probabilities, [256][K] level probabilities, [256][D] node ids
    const int64_t s0 = (int64_t)blockIdx.x * 256;
    const int cnt = (int)min((int64_t)256, n - s0);
    double *s_pp = s_rc, *s_lp = s_rc + 256 * D;
    int32_t *s_pn = reinterpret_cast<int32_t *>(s_lp + 256 * K);
    const int64_t s = s0 + threadIdx.x;
    if (threadIdx.x < cnt) {
        int sel = -1, lead = 0;
        int st = RD(status, s);
        double *pp = s_pp + threadIdx.x * D;
        double *lp = s_lp + threadIdx.x * K;
        int32_t *pn = s_pn + threadIdx.x * D;
        int len = levels;
        for (int d = 0; d < levels; ++d) {   // independent loads; the walk ended at the first node < 0
            const LevelRec r = RD(lvl, (size_t)d * n + s);
            pp[d] = r.prob;
            pn[d] = r.node;
            if (r.node < 0) len = min(len, d);
        }
        for (int d = len; d < D; ++d) pn[d] = -1;
        const double STAG_CONST *tfc = as_const(tf);
        for (int kk = 0; kk < K; ++kk) lp[kk] = 0.0;
        if (st == 0 && len == 0) st = 4;  // perc[0] on an empty list
        if (st == 0) {
            bool leading = true;
            for (int i = 0; i < len; ++i) {
                double p = pp[i];
                if (p == 2.0) {
                    p = i == 0 ? 1.0 : pp[i - 1];
                    pp[i] = p;
                    if (leading) ++lead;
                } else {
                    leading = false;
                }
            }
            for (int i = len; i < D; ++i) pp[i] = 0.0;
            double best = 0.0;
            for (int kk = 0; kk < K; ++kk) {
                const double STAG_CONST *row = tfc + (size_t)kk * (1 + n_tf);
                double acc = 0.0;
                if (len == n_tf) {
                    for (int i = 0; i < len; ++i) acc += RDK(row, 1 + i) * pp[i];
                } else if (len == 1) {
                    for (int i = 0; i < n_tf; ++i) acc += RDK(row, 1 + i) * pp[0];
                } else if (n_tf == 1) {
                    for (int i = 0; i < len; ++i) acc += RDK(row, 1) * pp[i];
                } else {
                    st = 3;
                    break;
                }
                const double p = sigmoid(acc + RDK(row, 0));
                lp[kk] = p;
                if (p > best) { best = p; sel = kk; }
            }
            if (st != 0) sel = -1;
        } else {
            for (int i = 0; i < D; ++i) pp[i] = 0.0;
        }
        WR(path_len, s, len);
        WR(n_lead_single, s, lead);
        WR(selected, s, sel);
        WR(status, s, st);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cnt * D; i += 256) {
        WR(path_prob, (size_t)s0 * D + i, s_pp[i]);
        WR(path_node, (size_t)s0 * D + i, s_pn[i]);
    }
    for (int i = threadIdx.x; i < cnt * K; i += 256) WR(level_prob, (size_t)s0 * K + i, s_lp[i]);
}

}  // namespace stagk
