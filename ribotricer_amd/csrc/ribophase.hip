// ribophase.hip -- C ABI of libribophase.so (see include/ribophase.h).
//
// Host side of the MI355X phase-score engine: argument checking, workspace carving,
// kernel selection and launches on the caller's stream.  No device allocation, no
// CPU compute path: if HIP is unusable every compute entry point fails.
#include <hip/hip_runtime.h>

#include <atomic>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <vector>
#include <algorithm>

#include "../../include/ribophase.h"
#include "rp_device.hpp"
#include "rp_tile.hpp"
#include <new>
#include "rp_format.hpp"
#include "rp_index.hpp"
#include "rp_bam.hpp"
#include "rp_wave.hpp"
#include "rp_coverage.hpp"
#include "rp_replay.hpp"

namespace {

thread_local char g_err[512] = "";
constexpr long long kAutoWaveNt = 2LL << 20;  // RP_ALGO_AUTO switches to the wave kernel below this

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define RP_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail(RP_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                           \
    } while (0)

int check_device(int device)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(RP_ERR_DEVICE, "no HIP device available (%s)",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    if (device < 0 || device >= n) return fail(RP_ERR_DEVICE, "device %d out of range [0,%d)", device, n);
    return RP_OK;
}

// Makes `device` current for the lifetime of the object and restores the calling thread's
// previous device afterwards: an entry point must not leave the caller (torch, another
// engine thread) on a different GPU than it came in with.
struct DeviceGuard {
    int rc = RP_OK;
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int device)
    {
        rc = check_device(device);
        if (rc != RP_OK) return;
        hipError_t e = hipGetDevice(&prev);
        if (e != hipSuccess) {
            rc = fail(RP_ERR_HIP, "hipGetDevice failed: %s", hipGetErrorString(e));
            return;
        }
        if (prev != device) {
            e = hipSetDevice(device);
            if (e != hipSuccess) {
                rc = fail(RP_ERR_HIP, "hipSetDevice(%d) failed: %s", device, hipGetErrorString(e));
                return;
            }
            switched = true;
        }
    }
    ~DeviceGuard()
    {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

#define RP_ON_DEVICE(device)       \
    DeviceGuard guard_((device));  \
    if (guard_.rc != RP_OK) return guard_.rc

// The tie replay (rp_device.hpp, replay_tie_wave) needs, per codon (a,b,c), exactly what the
// reference's Python + scipy make of it (statistics.py:75-90 and the first half of
// scipy.signal.coherence): norm = sqrt(pow(real,2) + pow(image,2)) with glibc's pow(), which
// is not x*x, then the three values of the segment's spectra.  They depend on the codon alone,
// so all codons with counts < 16 are tabulated HERE, once per device, with this host's libm --
// the libm the reference itself would run on (rp_replay.hpp: codon_terms).
constexpr int kMaxDevices = 64;
std::mutex g_tab_mutex;
bool g_tab_ready[kMaxDevices] = {};

void fill_codon_table(rp::CodonTerms *tab)
{
    constexpr int B = rp::kCodonTabBits, N = 1 << B;
    for (int a = 0; a < N; ++a)
        for (int b = 0; b < N; ++b)
            for (int c = 0; c < N; ++c) {
                const rpreplay::Terms t = rpreplay::codon_terms((double)a, (double)b, (double)c);
                rp::CodonTerms &e = tab[(a << (2 * B)) | (b << B) | c];
                e.pxx = t.pxx;
                e.pxr = t.pxr;
                e.pxi = t.pxi;
                e.pad = 0.0;
            }
}

int ensure_norm_table(int device)  // call with `device` current
{
    if (device < 0 || device >= kMaxDevices) return fail(RP_ERR_DEVICE, "device index %d beyond the %d this build tracks", device, kMaxDevices);
    std::lock_guard<std::mutex> lock(g_tab_mutex);
    if (g_tab_ready[device]) return RP_OK;
    constexpr int N = 1 << (3 * rp::kCodonTabBits);
    static rp::CodonTerms tab[N];
    fill_codon_table(tab);
    RP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(rp::rp_codon_tab), tab, sizeof(tab)));
    g_tab_ready[device] = true;
    return RP_OK;
}

int grid_for_waves(long long n_items, int waves_per_block)
{
    // memory-bound grid-stride launch: cap at 256 CUs x 8 workgroups
    long long blocks = (n_items + waves_per_block - 1) / waves_per_block;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    return (int)blocks;
}

rp::FilterParams make_filter(const rp_filter_params *f, const uint8_t *d_status)
{
    rp::FilterParams fp;
    memset(&fp, 0, sizeof(fp));
    if (f != nullptr && d_status != nullptr) {
        fp.phase_score_cutoff = f->phase_score_cutoff;
        fp.min_valid_codons_ratio = f->min_valid_codons_ratio;
        fp.min_density_over_orf = f->min_density_over_orf;
        fp.min_reads_per_codon = f->min_reads_per_codon;
        fp.min_valid_codons = f->min_valid_codons;
        fp.enabled = 1;
        fp.printed_only = (f->flags & RP_FILTER_PRINTED_ONLY) ? 1 : 0;
    }
    return fp;
}

struct Timing {
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool on = false;
};

bool known_algo(int algo) { return algo == RP_ALGO_AUTO || algo == RP_ALGO_WAVE || algo == RP_ALGO_TILE; }

// the part of a launch that depends on the offsets only: tile index + segment descriptors
// run `call` with the tile size as a compile-time constant (the two sizes of rp_tile.hpp)
#define RP_WITH_TILE(tile, call)                 \
    do {                                         \
        if ((tile) == rp::kTile) {               \
            constexpr int TILE = rp::kTile;      \
            call;                                \
        } else {                                 \
            constexpr int TILE = rp::kTileSmall; \
            call;                                \
        }                                        \
    } while (0)

std::atomic<int> g_measurement_tag{0};


// workgroups of the scoring launch (RP_TILES_PER_WG tiles each, rp_tile.hpp)
inline unsigned score_grid(long long n_tiles)
{
    return (unsigned)((n_tiles + RP_TILES_PER_WG - 1) / RP_TILES_PER_WG);
}

int launch_plan_kernels(const int64_t *d_offsets, int64_t n_orfs, const rp::TilePlan &plan, int tile,
                        const rp::TileWorkspace &ws, int *d_err, hipStream_t stream)
{
    const int block = 256;
    // head rows and descriptors are contiguous: gaps (empty ORFs, unused ids, unused slots) read as 0
    RP_HIP(hipMemsetAsync(ws.head, 0, rp::head_bytes(plan.total_nt, tile) + (size_t)ws.n_rec * sizeof(rp::seg_desc_t), stream));
    {
        const long long threads = n_orfs + 1;
        const int grid = (int)((threads + block - 1) / block);
        RP_WITH_TILE(tile, hipLaunchKernelGGL(rp::k_tile_index<TILE>, dim3(grid), dim3(block), 0, stream, d_offsets,
                                              (long long)n_orfs, plan, ws.tile_first, d_err));
        RP_HIP(hipGetLastError());
    }
    if (n_orfs > 0) {
        const int grid = (int)((n_orfs + block - 1) / block);
        RP_WITH_TILE(tile, hipLaunchKernelGGL(rp::k_tile_desc<TILE>, dim3(grid), dim3(block), 0, stream, d_offsets,
                                              (long long)n_orfs, plan, ws.tile_first, ws.head, ws.desc, d_err ? d_err + 1 : nullptr));
        RP_HIP(hipGetLastError());
    }
    {
        const int per_block = rp::kHeadBlock / 64;  // a wave per tile
        const int grid = (int)((plan.n_tiles + per_block - 1) / per_block);
        hipLaunchKernelGGL(rp::k_tile_head, dim3(grid), dim3(rp::kHeadBlock), 0, stream, plan.n_tiles, ws.tile_first, ws.head);
        RP_HIP(hipGetLastError());
    }
    return RP_OK;
}

}  // namespace

// A tile plan: everything the tile path derives from the offsets alone (today: the tile
// index).  `detect-orfs` scores ONE candidate index against many samples, so the plan is
// built -- and the offsets validated -- once per index instead of once per call.
struct rp_plan {
    int device;
    long long n_orfs, total_nt;
    int mis;         // (counts address / 4) % 4 the plan was built for
    int tile;        // positions per tile (rp::pick_tile of the index)
    int n_long;      // ORFs longer than rp::kLongWalk (counted up to about rp::kLongCountCap): sizes / skips the k_rewalk_long launch
    void *tables;    // device, caller-owned (inside d_plan_mem): tile index + segment descriptors
    int *err;        // device, first word of d_plan_mem
};

// A gather plan: the profile space of an index as pieces of the dense coverage
// (rp_pieces.hpp), built once per index from the interval table.
struct rp_gather_plan {
    int device;
    long long n_orfs, n_pieces, total_nt, coverage_len, n_tiles;
    int tile;
    rp::PiecePlanMem mem;  // device, caller-owned (inside d_plan_mem)
};

#ifndef RP_LONG_GRID
#define RP_LONG_GRID 512  // workgroups of k_rewalk_long (grid-stride over the queue of long too-close-to-call ORFs, usually empty)
#endif

namespace {

constexpr size_t kPlanHeader = 128;

rp::PiecePlan piece_plan_of(const rp_gather_plan *g)
{
    return rp::PiecePlan{g->mem.start, g->mem.base, g->mem.orf_piece, g->mem.tile_piece0, g->mem.tile_lo, g->mem.rows, g->n_pieces, g->coverage_len};
}

int score_impl(int device, const int32_t *d_counts, const int64_t *d_offsets, int64_t n_orfs,
               int64_t total_nt, double *d_phase, int32_t *d_valid, int64_t *d_read_count,
               int32_t *d_min_codon_cov, uint8_t *d_flags, uint8_t *d_status,
               const rp_filter_params *filter, void *d_workspace, size_t workspace_bytes, int algo,
               const rp_plan *plan_h, void *hip_stream, Timing *tm, const rp_gather_plan *gather = nullptr)
{
    if (n_orfs < 0 || total_nt < 0) return fail(RP_ERR_SIZE, "n_orfs=%lld total_nt=%lld must be >= 0", (long long)n_orfs, (long long)total_nt);
    if (total_nt >= (1ll << 40))  // (4 TiB of counts: rp::tile_of keeps 32 bits of position >> 8)
        return fail(RP_ERR_SIZE, "total_nt=%lld exceeds 2^40 positions per call", (long long)total_nt);
    if (!known_algo(algo)) return fail(RP_ERR_ARG, "unknown algo %d", algo);
    if (n_orfs > 0 && (!d_offsets || !d_phase || !d_valid || !d_read_count || !d_min_codon_cov || !d_flags))
        return fail(RP_ERR_NULL, "offsets and the five output arrays must be non-null");
    if (total_nt > 0 && !d_counts) return fail(RP_ERR_NULL, "d_counts is null but total_nt > 0");
    RP_ON_DEVICE(device);
    int rc = ensure_norm_table(device);
    if (rc != RP_OK) return rc;
    hipStream_t stream = (hipStream_t)hip_stream;
    if (tm && tm->on) RP_HIP(hipEventRecord(tm->ev[0], stream));
    if (n_orfs == 0) {
        if (tm && tm->on)
            for (int k = 1; k < 4; ++k) RP_HIP(hipEventRecord(tm->ev[k], stream));
        return RP_OK;
    }
    const rp::OrfOutputs out{d_phase, d_valid, d_read_count, d_min_codon_cov, d_flags, d_status};
    const rp::FilterParams fp = make_filter(filter, d_status);
    // AUTO: the flat-tile path, except for batches so small that its launches cost more than
    // the single wave-per-ORF launch (measured crossover ~3 M nt; scripts/bench_small.py)
    if (algo == RP_ALGO_AUTO) algo = (plan_h == nullptr && total_nt < kAutoWaveNt) ? RP_ALGO_WAVE : RP_ALGO_TILE;
    if (gather != nullptr) {  // d_counts is the dense coverage: the tile path stages through the pieces
        algo = RP_ALGO_TILE;
        if (gather->device != device || gather->n_orfs != n_orfs || gather->total_nt != total_nt)
            return fail(RP_ERR_ARG, "gather plan was built for device %d, %lld ORFs, %lld nt; called with device %d, %lld ORFs, %lld nt",
                        gather->device, gather->n_orfs, gather->total_nt, device, (long long)n_orfs, (long long)total_nt);
    }

    if (algo == RP_ALGO_WAVE) {
        if (tm && tm->on) RP_HIP(hipEventRecord(tm->ev[1], stream));
        const int grid = grid_for_waves(n_orfs, rp::kWaveBlock / rp::kWave);
        hipLaunchKernelGGL(rp::k_wave_score, dim3(grid), dim3(rp::kWaveBlock), 0, stream, d_counts,
                           d_offsets, (long long)n_orfs, out, fp);
        RP_HIP(hipGetLastError());
        if (tm && tm->on) {
            RP_HIP(hipEventRecord(tm->ev[2], stream));
            RP_HIP(hipEventRecord(tm->ev[3], stream));
        }
        return RP_OK;
    }

    // with a plan only the records live in the workspace
    const int tile = rp::pick_tile(n_orfs, total_nt);
    const size_t need = plan_h ? rp::record_bytes(n_orfs, total_nt, tile) : rp::workspace_bytes(n_orfs, total_nt, tile);
    if (!d_workspace || workspace_bytes < need)
        return fail(RP_ERR_WORKSPACE, "workspace of %zu bytes required, got %zu", need, workspace_bytes);
    if ((reinterpret_cast<uintptr_t>(d_workspace) & 15u) != 0)
        return fail(RP_ERR_WORKSPACE, "workspace must be 16-byte aligned");
    // RP_ALGO_TILE: plan tables (the caller's, or built here) -> scoring pass (segment records) -> per-ORF finish
    const rp::TilePlan plan = rp::make_tile_plan(n_orfs, total_nt, gather ? 0 : rp::counts_phase(d_counts), tile);
    if (plan_h != nullptr) {
        if (plan_h->device != device || plan_h->n_orfs != n_orfs || plan_h->total_nt != total_nt)
            return fail(RP_ERR_ARG, "plan was built for device %d, %lld ORFs, %lld nt; called with device %d, %lld ORFs, %lld nt",
                        plan_h->device, plan_h->n_orfs, plan_h->total_nt, device, (long long)n_orfs, (long long)total_nt);
        if (plan_h->mis != plan.mis)
            return fail(RP_ERR_ARG, "plan was built for counts at 16-byte phase %d, d_counts has phase %d", plan_h->mis, plan.mis);
    }
    const rp::TileWorkspace ws = rp::carve_workspace(d_workspace, plan_h ? plan_h->tables : nullptr, n_orfs, total_nt, tile);
    if (plan_h == nullptr) {
        // 1. tile index (first ORF starting at or after each tile boundary) + segment descriptors
        rc = launch_plan_kernels(d_offsets, n_orfs, plan, tile, ws, nullptr, stream);
        if (rc != RP_OK) return rc;
    }
    if (tm && tm->on) RP_HIP(hipEventRecord(tm->ev[1], stream));
    // 2. scoring pass over flat tiles: one record per (ORF, tile) segment
    const bool tagged = g_measurement_tag.load(std::memory_order_relaxed) != 0;  // (rp_measurement_tag: same code, second kernel name)
    if (gather != nullptr) {
        if (tagged)
            RP_WITH_TILE(tile, hipLaunchKernelGGL((rp::k_tile_score_probe<true, TILE>), dim3(score_grid(plan.n_tiles)), dim3(rp::kTileBlock), 0, stream,
                                                  d_counts, (long long)n_orfs, plan, ws, piece_plan_of(gather)));
        else
            RP_WITH_TILE(tile, hipLaunchKernelGGL((rp::k_tile_score<true, TILE>), dim3(score_grid(plan.n_tiles)), dim3(rp::kTileBlock), 0, stream,
                                                  d_counts, (long long)n_orfs, plan, ws, piece_plan_of(gather)));
    } else {
        if (tagged)
            RP_WITH_TILE(tile, hipLaunchKernelGGL((rp::k_tile_score_probe<false, TILE>), dim3(score_grid(plan.n_tiles)), dim3(rp::kTileBlock), 0, stream,
                                                  d_counts, (long long)n_orfs, plan, ws, rp::PiecePlan{}));
        else
            RP_WITH_TILE(tile, hipLaunchKernelGGL((rp::k_tile_score<false, TILE>), dim3(score_grid(plan.n_tiles)), dim3(rp::kTileBlock), 0, stream,
                                                  d_counts, (long long)n_orfs, plan, ws, rp::PiecePlan{}));
    }
    RP_HIP(hipGetLastError());
    if (tm && tm->on) RP_HIP(hipEventRecord(tm->ev[2], stream));
    // 3. one thread per ORF: add its records, score, filter, store (one one-wave workgroup per batch of 64 ORFs)
    {
        const int block = rp::kFinishBlock;
        const long long grid = (n_orfs + block - 1) / block;
        if (gather != nullptr)
            RP_WITH_TILE(tile, hipLaunchKernelGGL((rp::k_orf_finish<TILE, rp::CoverageSource>), dim3((unsigned)grid), dim3(block), 0, stream,
                                                  rp::CoverageSource{d_counts, piece_plan_of(gather)}, d_offsets, (long long)n_orfs, plan, ws, out, fp));
        else
            RP_WITH_TILE(tile, hipLaunchKernelGGL((rp::k_orf_finish<TILE, rp::CsrSource>), dim3((unsigned)grid), dim3(block), 0, stream,
                                                  rp::CsrSource{d_counts}, d_offsets, (long long)n_orfs, plan, ws, out, fp));
        RP_HIP(hipGetLastError());
        // 4. the long too-close-to-call ORFs it queued (usually none): a workgroup each.  A plan knows how many ORFs of
        //    its index are that long at all: none -> no launch (an empty 512-workgroup launch is 11 us, 3 % of the step of
        //    an eighth of the 11 M-ORF set), a few -> a grid of that many
        long long cap = rp::long_capacity(total_nt);
        if (plan_h != nullptr && plan_h->n_long < cap) cap = plan_h->n_long;
        if (total_nt > rp::kLongWalk && cap > 0) {
            const dim3 lgrid((unsigned)(cap < RP_LONG_GRID ? cap : RP_LONG_GRID));
            if (gather != nullptr)
                RP_WITH_TILE(tile, hipLaunchKernelGGL((rp::k_rewalk_long<TILE, rp::CoverageSource>), lgrid, dim3(rp::kLongBlock), 0, stream,
                                                      rp::CoverageSource{d_counts, piece_plan_of(gather)}, d_offsets, plan, ws, out, fp));
            else
                RP_WITH_TILE(tile, hipLaunchKernelGGL((rp::k_rewalk_long<TILE, rp::CsrSource>), lgrid, dim3(rp::kLongBlock), 0, stream,
                                                      rp::CsrSource{d_counts}, d_offsets, plan, ws, out, fp));
            RP_HIP(hipGetLastError());
        }
    }
    if (tm && tm->on) RP_HIP(hipEventRecord(tm->ev[3], stream));
    return RP_OK;
}

// score with HIP events around the phases: ms = {plan kernels, scoring kernel, finish, whole call}
template <typename Call>
int run_timed(int device, float ms[4], Call &&call)
{
    RP_ON_DEVICE(device);
    Timing tm;
    tm.on = true;
    for (int k = 0; k < 4; ++k) RP_HIP(hipEventCreate(&tm.ev[k]));
    int rc = call(&tm);
    if (rc == RP_OK) {
        hipError_t e = hipEventSynchronize(tm.ev[3]);
        if (e != hipSuccess) rc = fail(RP_ERR_HIP, "hipEventSynchronize: %s", hipGetErrorString(e));
    }
    if (rc == RP_OK) {
        (void)hipEventElapsedTime(&ms[0], tm.ev[0], tm.ev[1]);
        (void)hipEventElapsedTime(&ms[1], tm.ev[1], tm.ev[2]);
        (void)hipEventElapsedTime(&ms[2], tm.ev[2], tm.ev[3]);
        (void)hipEventElapsedTime(&ms[3], tm.ev[0], tm.ev[3]);
    }
    for (int k = 0; k < 4; ++k) (void)hipEventDestroy(tm.ev[k]);
    return rc;
}

}  // namespace

extern "C" {

#ifdef RP_STAMPS
// measurement builds only: read the phase stamps of k_tile_score (kStampSlots x 4 x 8 uint64)
int rp_debug_read_stamps(unsigned long long *out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(rp::rp_dbg_stamps), sizeof(unsigned long long) * rp::kStampSlots * 32) == hipSuccess ? 0 : -1;
}
#endif

const char *rp_version(void) { return RP_VERSION_STRING; }

const char *rp_last_error(void) { return g_err; }

const char *rp_status_string(int status)
{
    switch (status) {
        case RP_OK: return "ok";
        case RP_ERR_NULL: return "null pointer";
        case RP_ERR_SIZE: return "bad size";
        case RP_ERR_OFFSETS: return "bad CSR offsets";
        case RP_ERR_HIP: return "HIP runtime error";
        case RP_ERR_WORKSPACE: return "bad workspace";
        case RP_ERR_DEVICE: return "no such HIP device";
        case RP_ERR_COUNTS: return "count out of range";
        case RP_ERR_ARG: return "invalid argument";
        case RP_ERR_INDEX_COLUMNS: return "index line: unexpected number of columns";
        case RP_ERR_INDEX_COORD: return "index line: malformed coordinate";
        case RP_ERR_BAM: return "unreadable BAM file";
        case RP_ERR_INTERVALS: return "interval table cannot be planned";
        default: return "unknown status";
    }
}

int rp_device_count(int *n_devices)
{
    if (!n_devices) return fail(RP_ERR_NULL, "n_devices is null");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        *n_devices = 0;
        return fail(RP_ERR_DEVICE, "no HIP device available (%s)",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    *n_devices = n;
    return RP_OK;
}

int rp_filter_defaults(rp_filter_params *out)
{
    if (!out) return fail(RP_ERR_NULL, "out is null");
    out->phase_score_cutoff = 0.428571428571;  // const.py:20
    out->min_valid_codons_ratio = 0.0;         // const.py:35
    out->min_density_over_orf = 0.0;           // const.py:39
    out->min_reads_per_codon = 0.0;            // const.py:32
    out->min_valid_codons = 5;                 // const.py:27
    out->flags = 0;
    return RP_OK;
}

int rp_workspace_bytes(int64_t n_orfs, int64_t total_nt, int algo, size_t *bytes)
{
    if (!bytes) return fail(RP_ERR_NULL, "bytes is null");
    if (n_orfs < 0 || total_nt < 0) return fail(RP_ERR_SIZE, "negative size");
    if (!known_algo(algo)) return fail(RP_ERR_ARG, "unknown algo %d", algo);
    *bytes = algo == RP_ALGO_WAVE ? 0 : rp::workspace_bytes(n_orfs, total_nt, rp::pick_tile(n_orfs, total_nt));
    return RP_OK;
}

int rp_tile_positions(int64_t n_orfs, int64_t total_nt, int32_t *positions)
{
    if (!positions) return fail(RP_ERR_NULL, "positions is null");
    if (n_orfs < 0 || total_nt < 0) return fail(RP_ERR_SIZE, "negative size");
    *positions = rp::pick_tile(n_orfs, total_nt);
    return RP_OK;
}

int rp_plan_bytes(int64_t n_orfs, int64_t total_nt, size_t *bytes)
{
    if (!bytes) return fail(RP_ERR_NULL, "bytes is null");
    if (n_orfs < 0 || total_nt < 0) return fail(RP_ERR_SIZE, "negative size");
    *bytes = kPlanHeader + rp::plan_bytes(n_orfs, total_nt, rp::pick_tile(n_orfs, total_nt));
    return RP_OK;
}

int rp_plan_create_dev(int device, const int64_t *d_offsets, int64_t n_orfs, int64_t total_nt,
                       int counts_phase, void *d_plan_mem, size_t plan_bytes, void *hip_stream,
                       rp_plan **out)
{
    if (!out) return fail(RP_ERR_NULL, "out is null");
    *out = nullptr;
    if (n_orfs < 0 || total_nt < 0) return fail(RP_ERR_SIZE, "negative size");
    if (!d_offsets || !d_plan_mem) return fail(RP_ERR_NULL, "d_offsets / d_plan_mem is null");
    if (counts_phase < 0 || counts_phase > 3) return fail(RP_ERR_ARG, "counts_phase must be (address / 4) %% 4");
    size_t need = 0;
    int rc = rp_plan_bytes(n_orfs, total_nt, &need);
    if (rc != RP_OK) return rc;
    if (plan_bytes < need) return fail(RP_ERR_WORKSPACE, "plan memory of %zu bytes required, got %zu", need, plan_bytes);
    if ((reinterpret_cast<uintptr_t>(d_plan_mem) & 15u) != 0) return fail(RP_ERR_WORKSPACE, "plan memory must be 16-byte aligned");
    RP_ON_DEVICE(device);
    hipStream_t stream = (hipStream_t)hip_stream;
    const int tile = rp::pick_tile(n_orfs, total_nt);
    const rp::TilePlan plan = rp::make_tile_plan(n_orfs, total_nt, counts_phase, tile);
    int *d_err = reinterpret_cast<int *>(d_plan_mem);
    void *tables = reinterpret_cast<char *>(d_plan_mem) + kPlanHeader;
    rp::TileWorkspace ws = rp::carve_workspace(nullptr, tables, n_orfs, total_nt, tile);
    RP_HIP(hipMemsetAsync(d_err, 0, kPlanHeader, stream));
    rc = launch_plan_kernels(d_offsets, n_orfs, plan, tile, ws, d_err, stream);
    if (rc != RP_OK) return rc;
    int h_head[2] = {0, 0};  // {error bits, ORFs longer than kLongWalk}
    RP_HIP(hipMemcpyAsync(h_head, d_err, sizeof(h_head), hipMemcpyDeviceToHost, stream));
    RP_HIP(hipStreamSynchronize(stream));
    const int h_err = h_head[0];
    if (h_err != 0) return fail(RP_ERR_OFFSETS, "offsets must start at 0, be monotone and end at total_nt");
    rp_plan *h = new (std::nothrow) rp_plan;
    if (!h) return fail(RP_ERR_SIZE, "out of memory");
    h->device = device;
    h->n_orfs = n_orfs;
    h->total_nt = total_nt;
    h->mis = counts_phase;
    h->tile = tile;
    h->n_long = h_head[1];
    h->tables = tables;
    h->err = d_err;
    *out = h;
    return RP_OK;
}

void rp_plan_free(rp_plan *plan) { delete plan; }

int rp_phase_score_csr_dev(int device, const int32_t *d_counts, const int64_t *d_offsets,
                           int64_t n_orfs, int64_t total_nt, double *d_phase, int32_t *d_valid,
                           int64_t *d_read_count, int32_t *d_min_codon_cov, uint8_t *d_flags,
                           uint8_t *d_status, const rp_filter_params *filter, void *d_workspace,
                           size_t workspace_bytes, int algo, void *hip_stream)
{
    return score_impl(device, d_counts, d_offsets, n_orfs, total_nt, d_phase, d_valid, d_read_count,
                      d_min_codon_cov, d_flags, d_status, filter, d_workspace, workspace_bytes, algo,
                      nullptr, hip_stream, nullptr);
}

int rp_phase_score_csr_plan_dev(const rp_plan *plan, const int32_t *d_counts, const int64_t *d_offsets,
                                double *d_phase, int32_t *d_valid, int64_t *d_read_count,
                                int32_t *d_min_codon_cov, uint8_t *d_flags, uint8_t *d_status,
                                const rp_filter_params *filter, void *d_workspace,
                                size_t workspace_bytes, void *hip_stream)
{
    if (!plan) return fail(RP_ERR_NULL, "plan is null");
    return score_impl(plan->device, d_counts, d_offsets, plan->n_orfs, plan->total_nt, d_phase, d_valid,
                      d_read_count, d_min_codon_cov, d_flags, d_status, filter, d_workspace,
                      workspace_bytes, RP_ALGO_TILE, plan, hip_stream, nullptr);
}

int rp_phase_score_csr_dev_timed(int device, const int32_t *d_counts, const int64_t *d_offsets,
                                 int64_t n_orfs, int64_t total_nt, double *d_phase,
                                 int32_t *d_valid, int64_t *d_read_count,
                                 int32_t *d_min_codon_cov, uint8_t *d_flags, uint8_t *d_status,
                                 const rp_filter_params *filter, void *d_workspace,
                                 size_t workspace_bytes, int algo, const rp_plan *plan,
                                 void *hip_stream, float ms[4])
{
    if (!ms) return fail(RP_ERR_NULL, "ms is null");
    return run_timed(device, ms, [&](Timing *tm) {
        return score_impl(device, d_counts, d_offsets, n_orfs, total_nt, d_phase, d_valid, d_read_count,
                          d_min_codon_cov, d_flags, d_status, filter, d_workspace, workspace_bytes, algo,
                          plan, hip_stream, tm);
    });
}

int rp_phase_score_frames_dev(int device, const int32_t *d_counts, const int64_t *d_offsets,
                              int64_t n_orfs, double *d_frame_score, int32_t *d_frame_n,
                              int32_t *d_frame_m, void *hip_stream)
{
    if (n_orfs < 0) return fail(RP_ERR_SIZE, "n_orfs must be >= 0");
    if (n_orfs > 0 && (!d_offsets || !d_frame_score || !d_frame_n || !d_frame_m))
        return fail(RP_ERR_NULL, "offsets and the three output arrays must be non-null");
    RP_ON_DEVICE(device);
    if (n_orfs == 0) return RP_OK;
    const int grid = grid_for_waves(n_orfs, rp::kWaveBlock / rp::kWave);
    hipLaunchKernelGGL(rp::k_wave_frames, dim3(grid), dim3(rp::kWaveBlock), 0, (hipStream_t)hip_stream,
                       d_counts, d_offsets, (long long)n_orfs, d_frame_score, d_frame_n, d_frame_m);
    RP_HIP(hipGetLastError());
    return RP_OK;
}

int rp_phase_score_f64_csr_dev(int device, const double *d_values, const int64_t *d_offsets,
                               int64_t n_profiles, double *d_phase, int32_t *d_valid,
                               uint8_t *d_flags, void *hip_stream)
{
    if (n_profiles < 0) return fail(RP_ERR_SIZE, "n_profiles must be >= 0");
    if (n_profiles > 0 && (!d_offsets || !d_phase || !d_valid || !d_flags))
        return fail(RP_ERR_NULL, "offsets and the three output arrays must be non-null");
    RP_ON_DEVICE(device);
    if (n_profiles == 0) return RP_OK;
    const int grid = grid_for_waves(n_profiles, rp::kWaveBlock / rp::kWave);
    hipLaunchKernelGGL(rp::k_wave_score_f64in, dim3(grid), dim3(rp::kWaveBlock), 0,
                       (hipStream_t)hip_stream, d_values, d_offsets, (long long)n_profiles, d_phase,
                       d_valid, d_flags);
    RP_HIP(hipGetLastError());
    return RP_OK;
}

int rp_tie_replay_host(const int32_t *counts, const int64_t *offsets, int64_t n_profiles, double *phase, int32_t *valid)
{
    if (n_profiles < 0) return fail(RP_ERR_SIZE, "n_profiles must be >= 0");
    if (n_profiles > 0 && (!offsets || !phase || !valid)) return fail(RP_ERR_NULL, "offsets / phase / valid is null");
    for (int64_t i = 0; i < n_profiles; ++i) {
        const int64_t beg = offsets[i], len = offsets[i + 1] - beg;
        if (len < 0 || (len > 0 && !counts)) return fail(RP_ERR_OFFSETS, "profile %lld: bad offsets or null counts", (long long)i);
        rpreplay::replay_profile(counts + beg, len, &phase[i], &valid[i]);
    }
    return RP_OK;
}

int rp_tie_replay_f64_host(const double *values, const int64_t *offsets, int64_t n_profiles, double *phase, int32_t *valid)
{
    if (n_profiles < 0) return fail(RP_ERR_SIZE, "n_profiles must be >= 0");
    if (n_profiles > 0 && (!offsets || !phase || !valid)) return fail(RP_ERR_NULL, "offsets / phase / valid is null");
    for (int64_t i = 0; i < n_profiles; ++i) {
        const int64_t beg = offsets[i], len = offsets[i + 1] - beg;
        if (len < 0 || (len > 0 && !values)) return fail(RP_ERR_OFFSETS, "profile %lld: bad offsets or null values", (long long)i);
        rpreplay::replay_profile(values + beg, len, &phase[i], &valid[i]);
    }
    return RP_OK;
}

int rp_phase_score_csr_host(const int32_t *counts, const int64_t *offsets, int64_t n_orfs, double *phase, int32_t *valid,
                            int64_t *read_count, int32_t *min_codon_cov, uint8_t *flags, uint8_t *status,
                            const rp_filter_params *filter, int n_threads)
{
    if (n_orfs < 0) return fail(RP_ERR_SIZE, "n_orfs must be >= 0");
    if (n_orfs == 0) return RP_OK;
    if (!offsets || !phase || !valid || !read_count || !min_codon_cov || !flags) return fail(RP_ERR_NULL, "offsets and the five output arrays must be non-null");
    if (offsets[0] != 0) return fail(RP_ERR_OFFSETS, "offsets must start at 0");
    for (int64_t i = 0; i < n_orfs; ++i)
        if (offsets[i + 1] < offsets[i]) return fail(RP_ERR_OFFSETS, "offsets must be monotone (ORF %lld)", (long long)i);
    if (offsets[n_orfs] > 0 && !counts) return fail(RP_ERR_NULL, "counts is null but offsets[n] > 0");
    int threads = n_threads > 0 ? n_threads : rphost::usable_threads();
    if (threads < 1) threads = 1;
    if ((int64_t)threads > n_orfs) threads = (int)n_orfs;
    auto work = [&](int64_t first, int64_t last) {
        for (int64_t i = first; i < last; ++i) {
            const int32_t *v = counts + offsets[i];
            const int64_t len = offsets[i + 1] - offsets[i];
            rpreplay::replay_profile(v, len, &phase[i], &valid[i]);
            int64_t total = 0;
            int64_t mn = RP_MIN_CODON_COV_EMPTY;
            for (int64_t k = 0; k < len; k += 3) {  // collapse_coverage_to_codon, common.py:164-180 (the last codon may be partial)
                int64_t codon = v[k];
                if (k + 1 < len) codon += v[k + 1];
                if (k + 2 < len) codon += v[k + 2];
                total += codon;
                if (codon < mn) mn = codon;
            }
            read_count[i] = total;
            // (counts beyond 2^29: a codon sum can pass int32 -- the output saturates below the "empty" sentinel, the
            // predicate below sees the exact sum)
            min_codon_cov[i] = (int32_t)(len > 0 && mn > (int64_t)RP_MIN_CODON_COV_EMPTY - 1 ? (int64_t)RP_MIN_CODON_COV_EMPTY - 1 : mn);
            flags[i] = 0;
            if (status && filter) {
                const int64_t n_codons = len / 3 > 1 ? len / 3 : 1;  // detect_orfs.py:281
                const bool ok = phase[i] >= filter->phase_score_cutoff && valid[i] >= filter->min_valid_codons &&
                                (double)mn >= filter->min_reads_per_codon &&
                                (double)valid[i] / (double)n_codons >= filter->min_valid_codons_ratio &&
                                (double)total / (double)n_codons >= filter->min_density_over_orf;
                status[i] = ok ? 1 : 0;
            }
        }
    };
    if (threads == 1) {
        work(0, n_orfs);
    } else {  // contiguous ORF ranges balanced on nucleotides
        std::vector<std::thread> pool;
        const int64_t total_nt = offsets[n_orfs];
        int64_t first = 0;
        for (int t = 0; t < threads; ++t) {
            int64_t last = n_orfs;
            if (t + 1 < threads) {
                const int64_t target = total_nt / threads * (t + 1);
                last = std::lower_bound(offsets + first, offsets + n_orfs, target) - offsets;
                if (last < first) last = first;
            }
            pool.emplace_back(work, first, last);
            first = last;
        }
        for (auto &th : pool) th.join();
    }
    return RP_OK;
}

int rp_gather_profiles_dev(int device, const int32_t *d_coverage, int64_t coverage_len,
                           const int64_t *d_iv_start, const int32_t *d_iv_len,
                           const int64_t *d_orf_iv, const uint8_t *d_reverse,
                           const int64_t *d_offsets, int64_t n_orfs, int32_t *d_counts,
                           void *hip_stream)
{
    if (n_orfs < 0 || coverage_len < 0) return fail(RP_ERR_SIZE, "negative size");
    if (n_orfs > 0 && (!d_orf_iv || !d_reverse || !d_offsets))
        return fail(RP_ERR_NULL, "interval CSR, strand flags and offsets must be non-null");
    RP_ON_DEVICE(device);
    if (n_orfs == 0) return RP_OK;
    // one wave per batch of 64 ORFs
    const int grid = grid_for_waves((n_orfs + rp::kWave - 1) / rp::kWave, rp::kWaveBlock / rp::kWave);
    hipLaunchKernelGGL(rp::k_gather_profiles, dim3(grid), dim3(rp::kWaveBlock), 0, (hipStream_t)hip_stream,
                       d_coverage, (long long)coverage_len, d_iv_start, d_iv_len, d_orf_iv, d_reverse,
                       d_offsets, (long long)n_orfs, d_counts);
    RP_HIP(hipGetLastError());
    return RP_OK;
}

int rp_gather_plan_bytes(int64_t n_orfs, int64_t n_intervals, int64_t total_nt, size_t *bytes)
{
    if (!bytes) return fail(RP_ERR_NULL, "bytes is null");
    if (n_orfs < 0 || n_intervals < 0 || total_nt < 0) return fail(RP_ERR_SIZE, "negative size");
    const rp::TilePlan tp = rp::make_tile_plan(n_orfs, total_nt, 0, rp::pick_tile(n_orfs, total_nt));
    *bytes = kPlanHeader + rp::piece_plan_bytes(n_orfs, n_intervals, tp.n_tiles);
    return RP_OK;
}

int rp_gather_plan_create_dev(int device, const int64_t *d_iv_start, const int32_t *d_iv_len,
                              const int64_t *d_orf_iv, const uint8_t *d_reverse, const int64_t *d_offsets,
                              int64_t n_orfs, int64_t n_intervals, int64_t total_nt, int64_t coverage_len,
                              void *d_plan_mem, size_t plan_bytes, void *hip_stream, rp_gather_plan **out)
{
    if (!out) return fail(RP_ERR_NULL, "out is null");
    *out = nullptr;
    if (n_orfs < 0 || n_intervals < 0 || total_nt < 0 || coverage_len < 0) return fail(RP_ERR_SIZE, "negative size");
    if (!d_orf_iv || !d_offsets || !d_plan_mem || (n_orfs > 0 && !d_reverse) || (n_intervals > 0 && (!d_iv_start || !d_iv_len)))
        return fail(RP_ERR_NULL, "interval table, strand flags, offsets and plan memory must be non-null");
    size_t need = 0;
    int rc = rp_gather_plan_bytes(n_orfs, n_intervals, total_nt, &need);
    if (rc != RP_OK) return rc;
    if (plan_bytes < need) return fail(RP_ERR_WORKSPACE, "gather plan memory of %zu bytes required, got %zu", need, plan_bytes);
    if ((reinterpret_cast<uintptr_t>(d_plan_mem) & 15u) != 0) return fail(RP_ERR_WORKSPACE, "plan memory must be 16-byte aligned");
    RP_ON_DEVICE(device);
    hipStream_t stream = (hipStream_t)hip_stream;
    const int tile = rp::pick_tile(n_orfs, total_nt);
    const rp::TilePlan tp = rp::make_tile_plan(n_orfs, total_nt, 0, tile);
    int *d_err = reinterpret_cast<int *>(d_plan_mem);
    const rp::PiecePlanMem mem = rp::carve_piece_plan(reinterpret_cast<char *>(d_plan_mem) + kPlanHeader, n_orfs, n_intervals, tp.n_tiles);
    RP_HIP(hipMemsetAsync(d_err, 0, kPlanHeader, stream));
    {
        const int block = 256;
        const long long grid = (n_orfs + 1 + block - 1) / block;
        hipLaunchKernelGGL(rp::k_piece_build, dim3((unsigned)grid), dim3(block), 0, stream, d_iv_start, d_iv_len, d_orf_iv,
                           d_reverse, d_offsets, (long long)n_orfs, (long long)n_intervals, (long long)total_nt,
                           (long long)coverage_len, mem, d_err);
        RP_HIP(hipGetLastError());
    }
    int h_err = 0;
    RP_HIP(hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, stream));
    RP_HIP(hipStreamSynchronize(stream));
    if (h_err & 1) return fail(RP_ERR_OFFSETS, "the intervals of an ORF do not add up to its profile length (or orf_iv is not a CSR index of the intervals)");
    if (h_err & 2) return fail(RP_ERR_INTERVALS, "an interval is empty or reaches outside the coverage array: not plannable (rp_gather_profiles_dev reads such positions as 0)");
    RP_WITH_TILE(tile, hipLaunchKernelGGL((rp::k_chunk_rows<TILE, rp::kHalo>), dim3((unsigned)tp.n_tiles), dim3(rp::kRowBlock), 0, stream, mem,
                                          (long long)n_intervals, (long long)total_nt));
    RP_HIP(hipGetLastError());
    RP_HIP(hipStreamSynchronize(stream));
    rp_gather_plan *h = new (std::nothrow) rp_gather_plan;
    if (!h) return fail(RP_ERR_SIZE, "out of memory");
    h->device = device;
    h->n_orfs = n_orfs;
    h->n_pieces = n_intervals;
    h->total_nt = total_nt;
    h->coverage_len = coverage_len;
    h->n_tiles = tp.n_tiles;
    h->tile = tile;
    h->mem = mem;
    *out = h;
    return RP_OK;
}

void rp_gather_plan_free(rp_gather_plan *plan) { delete plan; }

int rp_gather_profiles_plan_dev(const rp_gather_plan *plan, const int32_t *d_coverage, int64_t coverage_len,
                                int32_t *d_counts, void *hip_stream)
{
    if (!plan) return fail(RP_ERR_NULL, "plan is null");
    if (coverage_len != plan->coverage_len) return fail(RP_ERR_ARG, "plan was built for a coverage of %lld positions, got %lld", plan->coverage_len, (long long)coverage_len);
    if (plan->total_nt == 0) return RP_OK;
    if (!d_coverage || !d_counts) return fail(RP_ERR_NULL, "d_coverage / d_counts is null");
    if ((reinterpret_cast<uintptr_t>(d_counts) & 15u) != 0) return fail(RP_ERR_ARG, "d_counts must be 16-byte aligned");
    RP_ON_DEVICE(plan->device);
    RP_WITH_TILE(plan->tile, hipLaunchKernelGGL((rp::k_tile_gather<TILE, rp::kHalo>), dim3((unsigned)plan->n_tiles), dim3(rp::kGatherTileBlock), 0,
                                                (hipStream_t)hip_stream, d_coverage, piece_plan_of(plan), plan->total_nt, d_counts));
    RP_HIP(hipGetLastError());
    return RP_OK;
}

int rp_gather_selected_plan_dev(const rp_gather_plan *plan, const int32_t *d_coverage, int64_t coverage_len,
                                const int64_t *d_chosen, int64_t n_chosen, const int64_t *d_out_offsets,
                                int32_t *d_counts, void *hip_stream)
{
    if (!plan) return fail(RP_ERR_NULL, "plan is null");
    if (coverage_len != plan->coverage_len) return fail(RP_ERR_ARG, "plan was built for a coverage of %lld positions, got %lld", plan->coverage_len, (long long)coverage_len);
    if (n_chosen < 0) return fail(RP_ERR_SIZE, "n_chosen=%lld must be >= 0", (long long)n_chosen);
    if (n_chosen == 0) return RP_OK;
    if (!d_coverage || !d_chosen || !d_out_offsets || !d_counts) return fail(RP_ERR_NULL, "d_coverage / d_chosen / d_out_offsets / d_counts is null");
    RP_ON_DEVICE(plan->device);
    const long long per_block = rp::kSelectedBlock / 64;
    const long long grid = (n_chosen + per_block - 1) / per_block;
    if (grid > 0x7fffffffLL) return fail(RP_ERR_SIZE, "too many ORFs chosen for one launch");
    hipLaunchKernelGGL(rp::k_gather_selected, dim3((unsigned)grid), dim3(rp::kSelectedBlock), 0, (hipStream_t)hip_stream, d_coverage,
                       piece_plan_of(plan), reinterpret_cast<const long long *>(d_chosen), (long long)n_chosen,
                       reinterpret_cast<const long long *>(d_out_offsets), d_counts);
    RP_HIP(hipGetLastError());
    return RP_OK;
}

int rp_phase_score_coverage_dev(int device, const int32_t *d_coverage, int64_t coverage_len,
                                const int64_t *d_offsets, int64_t n_orfs, int64_t total_nt, double *d_phase,
                                int32_t *d_valid, int64_t *d_read_count, int32_t *d_min_codon_cov,
                                uint8_t *d_flags, uint8_t *d_status, const rp_filter_params *filter,
                                void *d_workspace, size_t workspace_bytes, const rp_plan *plan,
                                const rp_gather_plan *gather, void *hip_stream, float *ms)
{
    if (!gather) return fail(RP_ERR_NULL, "gather plan is null");
    if (coverage_len != gather->coverage_len) return fail(RP_ERR_ARG, "gather plan was built for a coverage of %lld positions, got %lld", gather->coverage_len, (long long)coverage_len);
    if (ms)
        return run_timed(device, ms, [&](Timing *tm) {
            return score_impl(device, d_coverage, d_offsets, n_orfs, total_nt, d_phase, d_valid, d_read_count, d_min_codon_cov,
                              d_flags, d_status, filter, d_workspace, workspace_bytes, RP_ALGO_TILE, plan, hip_stream, tm, gather);
        });
    return score_impl(device, d_coverage, d_offsets, n_orfs, total_nt, d_phase, d_valid, d_read_count, d_min_codon_cov,
                      d_flags, d_status, filter, d_workspace, workspace_bytes, RP_ALGO_TILE, plan, hip_stream, nullptr, gather);
}

namespace {
// the block map of a compact coverage inside caller-owned device memory (rp_coverage_map_bytes): bits, rank, chunk sums
rp::BlockMap block_map_of(const void *mem, int64_t dense_len, int shift)
{
    if (!mem) return rp::BlockMap{nullptr, nullptr, 0, 0};
    const long long w = rp::map_words(dense_len, shift);
    const char *p = static_cast<const char *>(mem);
    return rp::BlockMap{reinterpret_cast<const unsigned long long *>(p), reinterpret_cast<const long long *>(p + (size_t)w * 8), w, shift};
}
}  // namespace

int rp_coverage_map_bytes(int64_t dense_len, int32_t block_positions, size_t *bytes)
{
    if (!bytes) return fail(RP_ERR_NULL, "bytes is null");
    if (dense_len < 0) return fail(RP_ERR_SIZE, "negative size");
    if (!rp::map_block_ok(block_positions)) return fail(RP_ERR_ARG, "block_positions must be a power of two from 1 to 64, got %d", (int)block_positions);
    *bytes = rp::map_bytes(dense_len, rp::map_shift(block_positions));
    return RP_OK;
}

int rp_coverage_map_create_dev(int device, int64_t *d_iv_start, const int32_t *d_iv_len, int64_t n_intervals, int64_t dense_len,
                               int32_t block_positions, void *d_map_mem, size_t map_bytes, void *hip_stream, int64_t *compact_len)
{
    if (!compact_len) return fail(RP_ERR_NULL, "compact_len is null");
    *compact_len = 0;
    if (n_intervals < 0 || dense_len < 0) return fail(RP_ERR_SIZE, "negative size");
    if (!rp::map_block_ok(block_positions)) return fail(RP_ERR_ARG, "block_positions must be a power of two from 1 to 64, got %d", (int)block_positions);
    const int shift = rp::map_shift(block_positions);
    if (!d_map_mem || (n_intervals > 0 && (!d_iv_start || !d_iv_len))) return fail(RP_ERR_NULL, "interval table and map memory must be non-null");
    if (map_bytes < rp::map_bytes(dense_len, shift)) return fail(RP_ERR_WORKSPACE, "block map memory of %zu bytes required, got %zu", rp::map_bytes(dense_len, shift), map_bytes);
    if ((reinterpret_cast<uintptr_t>(d_map_mem) & 7u) != 0) return fail(RP_ERR_WORKSPACE, "map memory must be 8-byte aligned");
    RP_ON_DEVICE(device);
    hipStream_t stream = (hipStream_t)hip_stream;
    const long long w = rp::map_words(dense_len, shift);
    const long long chunks = rp::map_chunks(w);
    char *p = static_cast<char *>(d_map_mem);
    unsigned long long *bits = reinterpret_cast<unsigned long long *>(p);
    long long *rank = reinterpret_cast<long long *>(p + (size_t)w * 8);
    long long *partial = reinterpret_cast<long long *>(p + (size_t)w * 8 + (size_t)(w + 1) * 8);
    int *d_err = reinterpret_cast<int *>(partial + chunks + 1);  // (the 256 spare bytes behind the chunk sums)
    RP_HIP(hipMemsetAsync(d_map_mem, 0, rp::map_bytes(dense_len, shift), stream));
    if (n_intervals > 0) {
        hipLaunchKernelGGL(rp::k_map_mark, dim3((unsigned)((n_intervals + 255) / 256)), dim3(256), 0, stream, d_iv_start, d_iv_len,
                           (long long)n_intervals, (long long)dense_len, shift, bits, d_err);
        RP_HIP(hipGetLastError());
    }
    if (w > 0) {
        hipLaunchKernelGGL(rp::k_map_chunk_sums, dim3((unsigned)chunks), dim3(256), 0, stream, bits, w, partial);
        hipLaunchKernelGGL(rp::k_map_scan_partials, dim3(1), dim3(1024), 0, stream, partial, chunks);
        hipLaunchKernelGGL(rp::k_map_rank, dim3((unsigned)chunks), dim3(256), 0, stream, bits, w, partial, rank);
        RP_HIP(hipGetLastError());
    }
    int h_err = 0;
    long long kept = 0;
    RP_HIP(hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, stream));
    RP_HIP(hipMemcpyAsync(&kept, rank + w, sizeof(long long), hipMemcpyDeviceToHost, stream));
    RP_HIP(hipStreamSynchronize(stream));
    if (h_err) return fail(RP_ERR_INTERVALS, "an interval is empty or reaches outside the dense layout: not mappable");
    if (n_intervals > 0) {
        hipLaunchKernelGGL(rp::k_map_remap, dim3((unsigned)((n_intervals + 255) / 256)), dim3(256), 0, stream, d_iv_start,
                           (long long)n_intervals, rp::BlockMap{bits, rank, w, shift});
        RP_HIP(hipGetLastError());
    }
    *compact_len = (int64_t)(kept << shift);
    return RP_OK;
}

int rp_coverage_map_remap_dev(int device, int64_t *d_positions, int64_t n_positions, const void *d_map_mem, int64_t dense_len,
                              int32_t block_positions, void *hip_stream)
{
    if (n_positions < 0 || dense_len < 0) return fail(RP_ERR_SIZE, "negative size");
    if (!rp::map_block_ok(block_positions)) return fail(RP_ERR_ARG, "block_positions must be a power of two from 1 to 64, got %d", (int)block_positions);
    if (n_positions == 0) return RP_OK;
    if (!d_positions || !d_map_mem) return fail(RP_ERR_NULL, "d_positions / d_map_mem is null");
    RP_ON_DEVICE(device);
    hipLaunchKernelGGL(rp::k_map_remap, dim3((unsigned)((n_positions + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, d_positions,
                       (long long)n_positions, block_map_of(d_map_mem, dense_len, rp::map_shift(block_positions)));
    RP_HIP(hipGetLastError());
    return RP_OK;
}

namespace {
// err bit 0: a sum passed RP_MAX_COUNT (fine when the caller asked to be told: it finishes those ORFs in float64);
// bit 1: negative, or past INT32_MAX -- not representable in the int32 coverage
int coverage_build_verdict(int h_err, int32_t *big_counts)
{
    if (h_err & 2) return fail(RP_ERR_COUNTS, "a P-site count is negative, or an accumulated count passed 2^31 - 1 (the coverage is int32)");
    if (h_err & 1) {
        if (!big_counts)
            return fail(RP_ERR_COUNTS, "an accumulated P-site count passed %d (pass big_counts to be told instead: "
                                       "rp_coverage_big_positions_dev then lists the positions)", RP_MAX_COUNT);
        *big_counts = 1;
    }
    return RP_OK;
}
}  // namespace

int rp_coverage_build_dev(int device, const int32_t *d_group, const int64_t *d_pos, const int32_t *d_count,
                          int64_t n_entries, const int64_t *d_group_start, const int64_t *d_group_lo,
                          const int64_t *d_group_hi, int32_t n_groups, int32_t *d_coverage,
                          int64_t coverage_len, void *hip_stream, int32_t *big_counts)
{
    if (big_counts) *big_counts = 0;
    if (n_entries < 0 || coverage_len < 0 || n_groups < 0) return fail(RP_ERR_SIZE, "negative size");
    if (n_entries > 0 && (!d_group || !d_pos || !d_count || !d_group_start || !d_group_lo || !d_group_hi))
        return fail(RP_ERR_NULL, "entry columns and group tables must be non-null");
    if (coverage_len > 0 && !d_coverage) return fail(RP_ERR_NULL, "d_coverage is null");
    RP_ON_DEVICE(device);
    if (n_entries == 0) return RP_OK;
    hipStream_t stream = (hipStream_t)hip_stream;
    int *d_err = nullptr;  // 4-byte scratch word for the range check (once per sample, not the hot path)
    RP_HIP(hipMalloc(&d_err, sizeof(int)));
    hipError_t e = hipMemsetAsync(d_err, 0, sizeof(int), stream);
    int h_err = 0;
    if (e == hipSuccess) {
        long long blocks = (n_entries + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(rp::k_coverage_build, dim3((unsigned)blocks), dim3(256), 0, stream, d_group, d_pos, d_count,
                           (long long)n_entries, d_group_start, d_group_lo, d_group_hi, (int)n_groups, d_coverage,
                           (long long)coverage_len, d_err);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_err);
    if (e != hipSuccess) return fail(RP_ERR_HIP, "coverage build: %s", hipGetErrorString(e));
    return coverage_build_verdict(h_err, big_counts);
}

int rp_coverage_build_rows_dev(int device, const uint8_t *d_strand, const int32_t *d_chrom, const int64_t *d_pos,
                               const int64_t *d_count, int64_t n_rows, const int32_t *d_lut, int32_t n_chroms,
                               const int64_t *d_group_start, const int64_t *d_group_lo, const int64_t *d_group_hi,
                               int32_t n_groups, int32_t *d_coverage, int64_t coverage_len, void *hip_stream,
                               int32_t *big_counts, const void *d_block_map, int64_t dense_len, int32_t block_positions)
{
    if (big_counts) *big_counts = 0;
    if (n_rows < 0 || coverage_len < 0 || n_groups < 0 || n_chroms < 0) return fail(RP_ERR_SIZE, "negative size");
    if (d_block_map && (dense_len <= 0 || !rp::map_block_ok(block_positions)))
        return fail(RP_ERR_ARG, "a block map needs the length of the dense layout and the block size it was built for");
    if (n_rows > 0 && (!d_strand || !d_chrom || !d_pos || !d_count || !d_lut || !d_group_start || !d_group_lo || !d_group_hi))
        return fail(RP_ERR_NULL, "row columns, lookup table and group tables must be non-null");
    if (coverage_len > 0 && !d_coverage) return fail(RP_ERR_NULL, "d_coverage is null");
    RP_ON_DEVICE(device);
    if (n_rows == 0) return RP_OK;
    hipStream_t stream = (hipStream_t)hip_stream;
    int *d_err = nullptr;  // 4-byte scratch word for the range check (once per sample, not the hot path)
    RP_HIP(hipMalloc(&d_err, sizeof(int)));
    hipError_t e = hipMemsetAsync(d_err, 0, sizeof(int), stream);
    int h_err = 0;
    if (e == hipSuccess) {
        long long blocks = (n_rows + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(rp::k_coverage_build_rows, dim3((unsigned)blocks), dim3(256), 0, stream, d_strand, d_chrom, d_pos, d_count,
                           (long long)n_rows, d_lut, (int)n_chroms, d_group_start, d_group_lo, d_group_hi, (int)n_groups, d_coverage,
                           (long long)coverage_len, d_err,
                           block_map_of(d_block_map, dense_len, d_block_map ? rp::map_shift(block_positions) : 0));
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_err);
    if (e != hipSuccess) return fail(RP_ERR_HIP, "coverage build: %s", hipGetErrorString(e));
    return coverage_build_verdict(h_err, big_counts);
}

int rp_coverage_big_positions_dev(int device, const int32_t *d_coverage, int64_t coverage_len, int64_t *d_positions,
                                  int64_t capacity, int64_t *n_found, void *hip_stream)
{
    if (!n_found) return fail(RP_ERR_NULL, "n_found is null");
    *n_found = 0;
    if (coverage_len < 0 || capacity < 0) return fail(RP_ERR_SIZE, "negative size");
    if (coverage_len > 0 && !d_coverage) return fail(RP_ERR_NULL, "d_coverage is null");
    if (capacity > 0 && !d_positions) return fail(RP_ERR_NULL, "d_positions is null but capacity > 0");
    RP_ON_DEVICE(device);
    if (coverage_len == 0) return RP_OK;
    hipStream_t stream = (hipStream_t)hip_stream;
    unsigned long long *d_found = nullptr;
    RP_HIP(hipMalloc(&d_found, sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(d_found, 0, sizeof(unsigned long long), stream);
    unsigned long long h_found = 0;
    if (e == hipSuccess) {
        long long blocks = (coverage_len + 255) / 256;
        if (blocks > 16384) blocks = 16384;
        hipLaunchKernelGGL(rp::k_big_positions, dim3((unsigned)blocks), dim3(256), 0, stream, d_coverage, (long long)coverage_len,
                           reinterpret_cast<long long *>(d_positions), (long long)capacity, d_found);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h_found, d_found, sizeof(h_found), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_found);
    if (e != hipSuccess) return fail(RP_ERR_HIP, "big positions: %s", hipGetErrorString(e));
    *n_found = (int64_t)h_found;
    return RP_OK;
}

int rp_metagene_dev(int device, const int32_t *d_counts, const int64_t *d_offsets, int64_t n_orfs,
                    int32_t max_positions, double *d_mean, double *d_sum, int32_t *d_seen, void *hip_stream)
{
    if (n_orfs < 0 || max_positions < 0) return fail(RP_ERR_SIZE, "negative size");
    if (max_positions > 0 && (!d_sum || !d_seen)) return fail(RP_ERR_NULL, "d_sum / d_seen is null");
    if (n_orfs > 0 && (!d_offsets || !d_mean)) return fail(RP_ERR_NULL, "d_offsets / d_mean is null");
    RP_ON_DEVICE(device);
    hipStream_t stream = (hipStream_t)hip_stream;
    if (n_orfs > 0) {
        hipLaunchKernelGGL(rp::k_metagene_means, dim3((unsigned)((n_orfs + 255) / 256)), dim3(256), 0, stream, d_counts,
                           d_offsets, (long long)n_orfs, d_mean);
        RP_HIP(hipGetLastError());
    }
    if (max_positions > 0) {
        hipLaunchKernelGGL(rp::k_metagene_sums, dim3((unsigned)((2 * max_positions + 63) / 64)), dim3(64), 0, stream,
                           d_counts, d_offsets, d_mean, (long long)n_orfs, (int)max_positions, d_sum, d_seen);
        RP_HIP(hipGetLastError());
    }
    return RP_OK;
}

int rp_metagene_host(const int32_t *counts, const int64_t *offsets, int64_t n_orfs, int32_t max_positions, double *mean,
                     double *sum, int32_t *seen)
{
    if (n_orfs < 0 || max_positions < 0) return fail(RP_ERR_SIZE, "negative size");
    if ((n_orfs > 0 && (!offsets || !mean)) || (max_positions > 0 && (!sum || !seen))) return fail(RP_ERR_NULL, "offsets / outputs must be non-null");
    for (int32_t t = 0; t < 2 * max_positions; ++t) {
        sum[t] = 0.0;
        seen[t] = 0;
    }
    // ORFs in index order; per slot the same float64 operations in the same order as rp::k_metagene_sums (and as pandas:
    // Series / mean, then Series.add(fill_value=0) ORF after ORF, metagene.py:213-228)
    for (int64_t i = 0; i < n_orfs; ++i) {
        const int64_t beg = offsets[i], len = offsets[i + 1] - beg;
        if (len < 0 || (len > 0 && !counts)) return fail(RP_ERR_OFFSETS, "profile %lld: bad offsets or null counts", (long long)i);
        int64_t s = 0;
        for (int64_t k = 0; k < len; ++k) s += counts[beg + k];
        const double m = len > 0 ? (double)s / (double)len : 0.0;
        mean[i] = m;
        if (!(m > 0.0)) continue;
        const int64_t n = len < max_positions ? len : max_positions;
        for (int64_t slot = 0; slot < n; ++slot) {
            sum[slot] = sum[slot] + (double)counts[beg + slot] / m;
            ++seen[slot];
            sum[max_positions + slot] = sum[max_positions + slot] + (double)counts[beg + len - 1 - slot] / m;
            ++seen[max_positions + slot];
        }
    }
    return RP_OK;
}

int rp_format_rows_host(const int32_t *counts, const int64_t *offsets, int64_t n_orfs,
                        const double *phase, const int32_t *valid, const int64_t *read_count,
                        const uint8_t *status, const char *head, const int64_t *head_off,
                        const char *tail, const int64_t *tail_off, int report_all, int64_t first,
                        char *out, size_t out_cap, int64_t *next, size_t *out_len)
{
    if (!next || !out_len) return fail(RP_ERR_NULL, "next / out_len is null");
    *next = first;
    *out_len = 0;
    if (n_orfs < 0 || first < 0 || first > n_orfs) return fail(RP_ERR_SIZE, "n_orfs=%lld first=%lld", (long long)n_orfs, (long long)first);
    if (first == n_orfs) return RP_OK;
    if (!offsets || !phase || !valid || !read_count || !status || !head_off || !tail_off || !out)
        return fail(RP_ERR_NULL, "format_rows: null array");
    if (!counts && offsets[n_orfs] > 0) return fail(RP_ERR_NULL, "counts is null but offsets[n] > 0");
    const rpfmt::RowInputs in{counts, offsets, phase, valid, read_count, status, head, head_off, tail, tail_off};
    size_t len = 0, need = 0;
    const long long nx = rpfmt::format_rows(in, n_orfs, report_all != 0, first, out, out_cap, &len, &need);
    *next = nx;
    *out_len = len;
    if (need > 0) {
        *out_len = need;
        return fail(RP_ERR_SIZE, "row of ORF %lld needs %zu bytes, buffer has %zu", nx, need, out_cap);
    }
    return RP_OK;
}

struct rp_index {
    rpidx::Index ix;
};

struct rp_bam {
    rpbam::Split sp;
};

int rp_bam_split_host(const char *path, int protocol, const int32_t *read_lengths, int32_t n_lengths, rp_bam **out)
{
    if (!out) return fail(RP_ERR_NULL, "out is null");
    *out = nullptr;
    if (!path) return fail(RP_ERR_NULL, "path is null");
    if (protocol != 0 && protocol != 1) return fail(RP_ERR_ARG, "protocol must be 0 (forward) or 1 (reverse)");
    if (n_lengths < 0 || (n_lengths > 0 && !read_lengths)) return fail(RP_ERR_ARG, "bad read_lengths");
    rp_bam *h = new (std::nothrow) rp_bam;
    if (!h) return fail(RP_ERR_SIZE, "out of memory");
    int rc = rpbam::kOk;
    try {
        rc = rpbam::split_bam(path, protocol, n_lengths > 0 ? read_lengths : nullptr, n_lengths, h->sp);
    } catch (const std::bad_alloc &) {
        delete h;
        return fail(RP_ERR_SIZE, "out of memory while reading the BAM file");
    }
    if (rc != rpbam::kOk) {
        const std::string msg = h->sp.error;
        delete h;
        return fail(RP_ERR_BAM, "%s: %s", path, msg.c_str());
    }
    *out = h;
    return RP_OK;
}

int rp_bam_view_host(const rp_bam *bam, rp_bam_view *view)
{
    if (!bam || !view) return fail(RP_ERR_NULL, "bam / view is null");
    const rpbam::Split &s = bam->sp;
    view->n_rows = (int64_t)s.pos.size();
    view->n_refs = (int64_t)s.ref_off.size() - 1;
    view->length = s.length.data();
    view->strand = s.strand.data();
    view->chrom = s.chrom.data();
    view->pos = s.pos.data();
    view->count = s.count.data();
    view->ref_names = s.ref_names.data();
    view->ref_off = s.ref_off.data();
    view->n_lengths = (int64_t)s.length_order.size();
    view->length_order = s.length_order.data();
    view->total = s.total;
    view->valid = s.valid;
    view->qcfail = s.qcfail;
    view->duplicate = s.duplicate;
    view->secondary = s.secondary;
    view->unmapped = s.unmapped;
    view->multi = s.multi;
    return RP_OK;
}

void rp_bam_free(rp_bam *bam) { delete bam; }

int rp_index_parse_host(const char *text, size_t len, int skip_header, rp_index **out, int64_t *error_line)
{
    if (!out) return fail(RP_ERR_NULL, "out is null");
    *out = nullptr;
    if (error_line) *error_line = 0;
    if (!text && len > 0) return fail(RP_ERR_NULL, "text is null but len > 0");
    rp_index *h = new (std::nothrow) rp_index;
    if (!h) return fail(RP_ERR_SIZE, "out of memory");
    int rc = rpidx::kOk;
    try {
        // RIBOPHASE_INDEX_THREADS: parser threads (default: up to 8; 1 = one sequential pass)
        const char *env = std::getenv("RIBOPHASE_INDEX_THREADS");
        rc = rpidx::parse(text, len, skip_header != 0, h->ix, env ? std::atoi(env) : 0);
    } catch (const std::bad_alloc &) {
        delete h;
        return fail(RP_ERR_SIZE, "out of memory while parsing the index");
    }
    if (rc != rpidx::kOk) {
        const long long line = h->ix.error_line;
        delete h;
        if (error_line) *error_line = line;
        if (rc == rpidx::kColumns)
            return fail(RP_ERR_INDEX_COLUMNS, "line %lld: unexpected number of columns found for index file", line);
        return fail(RP_ERR_INDEX_COORD, "line %lld: malformed coordinate field", line);
    }
    *out = h;
    return RP_OK;
}

int rp_index_view_host(const rp_index *index, rp_index_view *view)
{
    if (!index || !view) return fail(RP_ERR_NULL, "index / view is null");
    const rpidx::Index &ix = index->ix;
    view->n_orfs = (int64_t)ix.length.size();
    view->n_intervals = (int64_t)ix.iv_start.size();
    view->n_groups = (int64_t)ix.group_lo.size();
    view->orf_iv = ix.orf_iv.data();
    view->length = ix.length.data();
    view->group = ix.group.data();
    view->reverse = ix.reverse.data();
    view->iv_start = ix.iv_start.data();
    view->iv_end = ix.iv_end.data();
    view->group_names = ix.group_names.data();
    view->group_off = ix.group_off.data();
    view->group_lo = ix.group_lo.data();
    view->group_hi = ix.group_hi.data();
    view->head = ix.head.data();
    view->head_off = ix.head_off.data();
    view->tail = ix.tail.data();
    view->tail_off = ix.tail_off.data();
    return RP_OK;
}

void rp_index_free(rp_index *index) { delete index; }

int rp_interval_table_host(const int64_t *iv_start, const int64_t *iv_end, const int64_t *orf_iv, const int32_t *group,
                           const int64_t *length, int64_t n_orfs, int64_t n_intervals, const int64_t *group_start,
                           const int64_t *group_lo, int64_t n_groups, int64_t *out_iv_start, int32_t *out_iv_len,
                           int64_t *out_offsets)
{
    if (n_orfs < 0 || n_intervals < 0 || n_groups < 0) return fail(RP_ERR_SIZE, "negative size");
    if (!out_offsets || !orf_iv || (n_orfs > 0 && (!group || !length || !group_start || !group_lo)) ||
        (n_intervals > 0 && (!iv_start || !iv_end || !out_iv_start || !out_iv_len)))
        return fail(RP_ERR_NULL, "index arrays and outputs must be non-null");
    // ORF ranges side by side (22.8 M intervals: 85 ms on one core, most of it first-touch page faults of the outputs):
    // every thread writes its intervals and the running length sums of ITS range, the ranges' totals are scanned, and a
    // second sweep adds each range's base to its offsets
    out_offsets[0] = 0;
    int threads = rphost::usable_threads();
    if (threads > 32) threads = 32;
    if (n_orfs < 200000) threads = 1;
    struct Part {
        int64_t total = 0, bad = -1;
    };
    std::vector<Part> parts((size_t)threads);
    auto range = [&](int t) { return std::pair<int64_t, int64_t>{n_orfs * t / threads, n_orfs * (t + 1) / threads}; };
    auto fill = [&](int t) {
        const auto [a, b] = range(t);
        int64_t total = 0;
        for (int64_t i = a; i < b; ++i) {
            const int32_t g = group[i];
            const int64_t k0 = orf_iv[i], k1 = orf_iv[i + 1];
            if (g < 0 || g >= n_groups || k0 < 0 || k1 < k0 || k1 > n_intervals) {
                parts[(size_t)t].bad = i;
                return;
            }
            const int64_t shift = group_start[g] - group_lo[g];
            for (int64_t k = k0; k < k1; ++k) {
                out_iv_start[k] = iv_start[k] + shift;
                out_iv_len[k] = (int32_t)(iv_end[k] - iv_start[k] + 1);
            }
            total += length[i];
            out_offsets[i + 1] = total;
        }
        parts[(size_t)t].total = total;
    };
    auto rebase = [&](int t, int64_t base) {
        const auto [a, b] = range(t);
        for (int64_t i = a; i < b; ++i) out_offsets[i + 1] += base;
    };
    auto run = [&](auto &&fn) {
        if (threads == 1) {
            fn(0);
            return;
        }
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(fn, t);
        for (auto &th : pool) th.join();
    };
    run(fill);
    for (int t = 0; t < threads; ++t) {
        const int64_t i = parts[(size_t)t].bad;
        if (i >= 0)
            return fail(RP_ERR_ARG, "ORF %lld: group %d / interval range [%lld, %lld) out of range", (long long)i, (int)group[i],
                        (long long)orf_iv[i], (long long)orf_iv[i + 1]);
    }
    std::vector<int64_t> base((size_t)threads, 0);
    for (int t = 1; t < threads; ++t) base[(size_t)t] = base[(size_t)t - 1] + parts[(size_t)t - 1].total;
    if (threads > 1) run([&](int t) { if (base[(size_t)t] != 0) rebase(t, base[(size_t)t]); });
    return RP_OK;
}

int rp_gather_profiles_host(const int64_t *keys, const int64_t *vals, int64_t n_keys, const int64_t *iv_start,
                            const int64_t *iv_end, const int64_t *orf_iv, const int32_t *group, const uint8_t *reverse,
                            const int64_t *offsets, int64_t n_orfs, int32_t *counts, int n_threads)
{
    if (n_orfs < 0 || n_keys < 0) return fail(RP_ERR_SIZE, "negative size");
    if (n_orfs == 0) return RP_OK;
    if (!orf_iv || !group || !reverse || !offsets) return fail(RP_ERR_NULL, "index arrays and offsets must be non-null");
    if (n_keys > 0 && (!keys || !vals)) return fail(RP_ERR_NULL, "keys / vals is null");
    if (offsets[n_orfs] > 0 && (!counts || !iv_start || !iv_end)) return fail(RP_ERR_NULL, "counts / intervals is null");
    int threads = n_threads > 0 ? n_threads : rphost::usable_threads();
    if (threads > 64) threads = 64;
    if (n_orfs < 4096 || threads < 1) threads = 1;
    std::vector<int64_t> bad((size_t)threads, -1);
    std::vector<int> why((size_t)threads, 0);
    auto work = [&](int t) {
        const int64_t a = n_orfs * t / threads, b = n_orfs * (t + 1) / threads;
        for (int64_t i = a; i < b; ++i) {
            const int64_t beg = offsets[i], len = offsets[i + 1] - beg;
            const int64_t k0 = orf_iv[i], k1 = orf_iv[i + 1];
            const int64_t g = group[i];
            int64_t total = 0;
            for (int64_t k = k0; k < k1; ++k) total += iv_end[k] - iv_start[k] + 1;
            if (len < 0 || g < 0 || k1 < k0 || total != len) {
                bad[(size_t)t] = i;
                why[(size_t)t] = RP_ERR_OFFSETS;
                return;
            }
            int32_t *dst = counts + beg;
            std::memset(dst, 0, (size_t)len * sizeof(int32_t));  // a position that is not a key counts 0 (detect_orfs.py:176-187)
            const bool rev = reverse[i] != 0;
            int64_t asc = 0;  // ascending position inside the ORF of the interval's first nucleotide
            for (int64_t k = k0; k < k1; ++k) {
                const int64_t s = iv_start[k], e = iv_end[k];
                // (a leader that reaches below position 1, metagene.py:128-143, or a position past 2^40 is never a key:
                // the caller's keys hold 40 bits of position)
                const int64_t s_key = s < 0 ? 0 : s, e_key = e >= ((int64_t)1 << 40) ? ((int64_t)1 << 40) - 1 : e;
                if (s_key > e_key) {
                    asc += e - s + 1;
                    continue;
                }
                const int64_t lo_key = (g << 40) | s_key, hi_key = (g << 40) | e_key;
                const int64_t *p = std::lower_bound(keys, keys + n_keys, lo_key);
                for (; p < keys + n_keys && *p <= hi_key; ++p) {
                    const int64_t v = vals[p - keys];
                    if (v < 0 || v > INT32_MAX) {
                        bad[(size_t)t] = i;
                        why[(size_t)t] = RP_ERR_COUNTS;
                        return;
                    }
                    const int64_t at = asc + ((*p & (((int64_t)1 << 40) - 1)) - s);
                    dst[rev ? len - 1 - at : at] = (int32_t)v;  // '-' strand: the profile runs 5'->3' (detect_orfs.py:201-202)
                }
                asc += e - s + 1;
            }
        }
    };
    if (threads == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(work, t);
        for (auto &th : pool) th.join();
    }
    for (int t = 0; t < threads; ++t)
        if (bad[(size_t)t] >= 0) {
            if (why[(size_t)t] == RP_ERR_COUNTS)
                return fail(RP_ERR_COUNTS, "ORF %lld: a P-site count is negative or passes 2^31 - 1", (long long)bad[(size_t)t]);
            return fail(RP_ERR_OFFSETS, "ORF %lld: its intervals do not add up to its profile length", (long long)bad[(size_t)t]);
        }
    return RP_OK;
}

int rp_select_profiles_host(const uint8_t *keep, const int64_t *lengths, int64_t n_orfs, int64_t *chosen, int64_t *chosen_off,
                            int64_t *offsets, int64_t *n_chosen)
{
    if (!n_chosen || !offsets || !chosen_off) return fail(RP_ERR_NULL, "n_chosen, offsets and chosen_off must be non-null");
    *n_chosen = 0;
    if (n_orfs < 0) return fail(RP_ERR_SIZE, "negative size");
    if (n_orfs > 0 && (!keep || !lengths || !chosen)) return fail(RP_ERR_NULL, "keep, lengths and chosen must be non-null");
    int64_t k = 0, at = 0;
    offsets[0] = 0;
    chosen_off[0] = 0;
    for (int64_t i = 0; i < n_orfs; ++i) {
        if (keep[i]) {
            if (lengths[i] < 0) return fail(RP_ERR_ARG, "ORF %lld has a negative length", (long long)i);
            chosen[k] = i;
            at += lengths[i];
            chosen_off[++k] = at;
        }
        offsets[i + 1] = at;
    }
    *n_chosen = k;
    return RP_OK;
}

int rp_coverage_windows_host(const int64_t *iv_start, const int32_t *iv_len, int64_t n_intervals, int32_t gap_shift,
                             int64_t *win_start, int64_t *win_len, int64_t *win_base, int64_t capacity, int64_t *n_windows,
                             int64_t *total, int64_t *out_iv_start)
{
    if (!n_windows || !total) return fail(RP_ERR_NULL, "n_windows and total must be non-null");
    *n_windows = 0;
    *total = 0;
    if (n_intervals < 0 || capacity < 0) return fail(RP_ERR_SIZE, "negative size");
    if (gap_shift < 4 || gap_shift > 40) return fail(RP_ERR_ARG, "gap_shift must lie in [4, 40], got %d", (int)gap_shift);
    if (n_intervals == 0) return RP_OK;
    if (!iv_start || !iv_len || (capacity > 0 && (!win_start || !win_len || !win_base))) return fail(RP_ERR_NULL, "intervals and window arrays must be non-null");
    // pass 1: the extent (and the sanity of every interval)
    int64_t lo = INT64_MAX, hi = INT64_MIN;
    for (int64_t k = 0; k < n_intervals; ++k) {
        const int64_t s = iv_start[k], n = iv_len[k];
        if (n <= 0 || s < 0) return fail(RP_ERR_INTERVALS, "interval %lld is empty or starts below 0", (long long)k);
        if (s < lo) lo = s;
        if (s + n > hi) hi = s + n;
    }
    // pass 2: per block of 2^gap_shift positions, the lowest start and the highest end of the intervals that START in it.
    // Two intervals of one block start less than the gap apart: one window.  Blocks are then walked in order.
    const int64_t b0 = lo >> gap_shift;
    const int64_t n_blocks = ((hi - 1) >> gap_shift) - b0 + 1;
    std::vector<int64_t> first((size_t)n_blocks, INT64_MAX), last((size_t)n_blocks, INT64_MIN);
    for (int64_t k = 0; k < n_intervals; ++k) {
        const int64_t s = iv_start[k], e = s + iv_len[k];
        const size_t b = (size_t)((s >> gap_shift) - b0);
        if (s < first[b]) first[b] = s;
        if (e > last[b]) last[b] = e;
    }
    const int64_t gap = (int64_t)1 << gap_shift;
    std::vector<int64_t> block_window((size_t)n_blocks, -1);
    int64_t w = -1, reach = INT64_MIN, base = 0;
    auto close = [&]() {  // window w ends at `reach`
        if (w < 0) return;
        if (w < capacity) {
            win_len[w] = (reach - win_start[w] + 15) / 16 * 16;
            win_base[w] = base;
            base += win_len[w];
        }
    };
    int64_t start_w = 0;
    for (int64_t b = 0; b < n_blocks; ++b) {
        if (first[(size_t)b] == INT64_MAX) continue;
        if (w < 0 || first[(size_t)b] > reach + gap) {  // (sharding.coverage_windows: s > reach + gap opens a window)
            close();
            ++w;
            start_w = first[(size_t)b] / 16 * 16;
            if (w < capacity) win_start[w] = start_w;
            reach = last[(size_t)b];
        } else if (last[(size_t)b] > reach) {
            reach = last[(size_t)b];
        }
        block_window[(size_t)b] = w;
    }
    close();
    *n_windows = w + 1;
    if (w + 1 > capacity) return fail(RP_ERR_SIZE, "%lld windows, room for %lld", (long long)(w + 1), (long long)capacity);
    *total = base;
    if (out_iv_start) {
        for (int64_t k = 0; k < n_intervals; ++k) {
            const int64_t s = iv_start[k];
            const int64_t win = block_window[(size_t)((s >> gap_shift) - b0)];
            out_iv_start[k] = s - win_start[win] + win_base[win];
        }
    }
    return RP_OK;
}

int rp_format_double_repr(double value, char *buf) { return buf ? rpfmt::double_repr(value, buf) : 0; }

size_t rp_format_int_list(const int32_t *values, int64_t n, char *out)
{
    return (out && (values || n <= 0)) ? rpfmt::int_list_str(values, n, out) : 0;
}

int rp_wig_pack_host(const uint8_t *strand, const int32_t *chrom, const int64_t *pos, const int64_t *count, int64_t n_rows,
                     int32_t strand_code, const int32_t *rank_of_chrom, int32_t n_chroms, uint64_t *packed, int64_t *n_packed)
{
    if (!n_packed) return fail(RP_ERR_NULL, "n_packed is null");
    *n_packed = 0;
    if (n_rows < 0 || n_chroms < 0) return fail(RP_ERR_SIZE, "negative size");
    if (n_rows > 0 && (!strand || !chrom || !pos || !count || !rank_of_chrom || !packed)) return fail(RP_ERR_NULL, "columns, ranks and output must be non-null");
    if (n_chroms >= 1024) return fail(RP_ERR_ARG, "%d chromosome names do not fit the 10 rank bits of a packed WIG key", (int)n_chroms);
    int threads = rphost::usable_threads();
    if (threads > 32) threads = 32;
    if (n_rows < 1000000) threads = 1;
    std::vector<int64_t> kept((size_t)threads + 1, 0);
    std::vector<int> bad((size_t)threads, 0);
    auto range = [&](int t) { return std::pair<int64_t, int64_t>{n_rows * t / threads, n_rows * (t + 1) / threads}; };
    auto run = [&](auto &&fn) {
        if (threads == 1) {
            fn(0);
            return;
        }
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(fn, t);
        for (auto &th : pool) th.join();
    };
    run([&](int t) {  // how many rows of its range every thread keeps
        const auto [a, b] = range(t);
        int64_t k = 0;
        for (int64_t i = a; i < b; ++i) k += strand[i] == strand_code;
        kept[(size_t)t + 1] = k;
    });
    for (int t = 0; t < threads; ++t) kept[(size_t)t + 1] += kept[(size_t)t];
    run([&](int t) {
        const auto [a, b] = range(t);
        uint64_t *out = packed + kept[(size_t)t];
        for (int64_t i = a; i < b; ++i) {
            if (strand[i] != strand_code) continue;
            const int32_t c = chrom[i];
            const int64_t p = pos[i], v = count[i];
            if (c < 0 || c >= n_chroms || p < 0 || p >= (1ll << 32) || v < 0 || v >= (1ll << 22)) {
                bad[(size_t)t] = 1;
                return;
            }
            *out++ = ((uint64_t)(uint32_t)rank_of_chrom[c] << 54) | ((uint64_t)p << 22) | (uint64_t)v;
        }
    });
    for (int t = 0; t < threads; ++t)
        if (bad[(size_t)t]) return fail(RP_ERR_ARG, "a row does not fit a packed WIG key (position outside [0, 2^32), count outside [0, 2^22) or an unknown chromosome code)");
    *n_packed = kept[(size_t)threads];
    return RP_OK;
}

size_t rp_wig_render_host(const uint64_t *sorted_words, int64_t lo, int64_t hi, const char *names, const int64_t *name_off, char *out)
{
    if (!sorted_words || !names || !name_off || !out || lo < 0 || hi <= lo) return 0;
    char *o = out;
    // a chromosome's header goes in front of its first position: the range's first word opens one only at the very
    // start of the array or when the word before it belongs to another chromosome
    long long prev_rank = lo > 0 ? (long long)(sorted_words[lo - 1] >> 54) : -1;
    int64_t i = lo;
    while (i < hi) {
        const uint64_t key = sorted_words[i] >> 22;
        int64_t total = 0;
        for (; i < hi && (sorted_words[i] >> 22) == key; ++i) total += (int64_t)(sorted_words[i] & 0x3fffffull);
        const long long rank = (long long)(key >> 32);
        if (rank != prev_rank) {
            const int64_t a = name_off[rank], b = name_off[rank + 1];
            // (a chromosome named "" -- it sorts first -- gets no header: the reference's loop starts from cur_chrom = "",
            // detect_orfs.py:340-345)
            if (b > a) {
                static const char head[] = "variableStep chrom=";
                std::memcpy(o, head, sizeof(head) - 1);
                o += sizeof(head) - 1;
                std::memcpy(o, names + a, (size_t)(b - a));
                o += b - a;
                *o++ = '\n';
            }
            prev_rank = rank;
        }
        o += rpfmt::int_str((int64_t)(key & 0xffffffffull), o);
        *o++ = '\t';
        o += rpfmt::int_str(total, o);
        *o++ = '\n';
    }
    return (size_t)(o - out);
}

int rp_measurement_tag(int on)
{
    return g_measurement_tag.exchange(on ? 1 : 0, std::memory_order_relaxed);
}

size_t rp_format_wig_rows_host(const int64_t *pos, const int64_t *count, int64_t n, char *out)
{
    return (out && n > 0 && pos && count) ? rpfmt::wig_rows_str(pos, count, n, out) : 0;
}

int rp_validate_csr_dev(int device, const int32_t *d_counts, const int64_t *d_offsets,
                        int64_t n_orfs, int64_t total_nt, void *hip_stream)
{
    if (n_orfs < 0 || total_nt < 0) return fail(RP_ERR_SIZE, "negative size");
    if (!d_offsets) return fail(RP_ERR_NULL, "d_offsets is null");
    if (total_nt > 0 && !d_counts) return fail(RP_ERR_NULL, "d_counts is null but total_nt > 0");
    RP_ON_DEVICE(device);
    hipStream_t stream = (hipStream_t)hip_stream;
    int *d_err = nullptr;
    // validation is a debugging aid, not the hot path: it owns a 4-byte scratch word
    RP_HIP(hipMalloc(&d_err, sizeof(int)));
    hipError_t e = hipMemsetAsync(d_err, 0, sizeof(int), stream);
    int h_err = 0;
    if (e == hipSuccess) {
        long long work = total_nt > n_orfs ? total_nt : n_orfs + 1;
        long long blocks = (work + 255) / 256;
        if (blocks > 4096) blocks = 4096;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(rp::k_validate, dim3((unsigned)blocks), dim3(256), 0, stream, d_counts,
                           d_offsets, (long long)n_orfs, (long long)total_nt, d_err);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&h_err, d_err, sizeof(int), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d_err);
    if (e != hipSuccess) return fail(RP_ERR_HIP, "validate: %s", hipGetErrorString(e));
    if (h_err & 1) return fail(RP_ERR_OFFSETS, "offsets must start at 0, be monotone and end at total_nt");
    if (h_err & 2) return fail(RP_ERR_COUNTS, "counts must lie in [0, %d]", RP_MAX_COUNT);
    return RP_OK;
}

}  // extern "C"

#ifdef RP_REWALK_STAMPS  // (timing-experiment builds only: scripts/ab_finish_tail.py)
extern "C" int rp_debug_rewalk_stamps(unsigned long long *out16, int reset)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(rp::g_rewalk_stamps), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(rp::g_rewalk_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
extern "C" int rp_debug_replay_stamps(unsigned long long *out8, int reset)
{
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(rp::g_replay_stamps), 8 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(rp::g_replay_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

