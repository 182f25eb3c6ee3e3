// Internal declarations shared by the HIP translation units of libss_hip.so.
// Not part of the C-ABI (include/ss_hip.h is).
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "ss_hip.h"

namespace sship {

// Column pitch of the device copy is a multiple of this many ELEMENTS so that every
// wave-instruction of the sweep (64 lanes x 16 B) stays inside one column and the
// zero padding contributes exact zeros to the dot products.
constexpr uint32_t kRowPad = 256;
// Columns are padded (zero-filled) to a multiple of this so the sweep needs no tail code
// and the MFMA GEMM of the batched path sees whole 128-column tiles.
constexpr uint32_t kColPad = 256;
// DevState::status raised by the lookahead engine when the tolerance is too tight for Gram-form
// correlations (never leaves the library: the host re-runs the solve in residual form).
constexpr uint32_t kStatusRetryResidual = 100;
// DevState::status raised by the early form of the speculative engine when its first launch used more columns
// outside the prefetched ones than two fill-in passes fetch (the host re-runs the solve in the plain form).
constexpr uint32_t kStatusRetryPlain = 101;
// DevState::status raised by a step-length scan that met a tie stall (DevState::tie_stall) when the host asked for an early
// exit: the signal is re-run in the reference-order engine (never leaves the library).
constexpr uint32_t kStatusTieRerun = 102;
// subset form of the batched Gram form (subbatch.hip): columns per subset, positions, breakpoints a signal may log
constexpr uint32_t kSbS = 448, kSbRows = 72, kSbLog = 80;
constexpr uint32_t kStatusSubsetDecline = 110;      // left the form's common path: solved again in the lock-step form
constexpr uint32_t kStatusSubsetFail = 111;         // a column outside the subset would have changed a breakpoint
// Gram form is used while tolerance >= guard * ||A^T y||_inf: 2^-14 in fp32, 2^-42 in fp64 (eps x ~1000)
constexpr double kGramGuard = 1.0 / 16384.0;
constexpr double kGramGuard64 = 1.0 / 4398046511104.0;
// Hard cap of the active-set capacity (workspace is 2 * Kcap^2 elements).
constexpr uint32_t kKcapLimit = 4096;
// Upper bound of workgroups any sweep variant launches (size of the partial-max arrays).
constexpr uint32_t kMaxSweepBlocks = 1024;
// Upper bound of workgroups of the gamma scan.
constexpr uint32_t kMaxScanBlocks = 256;

// Device-resident solver state, one per in-flight signal.  Written by single-workgroup
// kernels, read by everything else; lives in global memory (L2-resident).
struct DevState {
    // ---- line 0: written only by single-workgroup code, read by every kernel ----------
    uint32_t done;        // 1 once the solve has terminated; every kernel becomes a no-op
    uint32_t status;      // ss_hip_status raised on the device (capacity overflow)
    uint32_t iter;        // homotopy iterations performed (ss::homotopy_report::iter)
    uint32_t K;           // current support size |Gamma|
    uint32_t ntouched;    // columns that were ever in the support (x may be non-zero there)
    uint32_t idx;         // column toggled by the current iteration
    uint32_t rank;        // its rank in the sorted support
    uint32_t added;       // 1 insert, 0 remove
    uint32_t cur;         // which of the two inverse / support buffers is current
    uint32_t done_round;  // round (1-based) in which `done` was raised
    double   c_inf;       // lambda = ||c||_inf (ss::homotopy_report::solution_error)
    double   gamma;       // step length of the current iteration
    // lookahead engine: Gram-column cache
    uint32_t need_sweep;  // 1 when the column being inserted has no cached Gram column yet
    uint32_t cache_used;  // cache slots handed out so far
    uint32_t nsweeps;     // lookahead sweeps that did work in this solve
    uint32_t seq;         // k_la_iter launches executed so far (mirrored to hflags[0])
    uint32_t nmiss;       // iterations that had to wait for a lookahead sweep (mirrored to hflags[2])
    uint32_t bar_rounds;  // k_la_iter launches that went through the grid barrier (bar_count = this * grid)
    // speculative ("solo") form of the resident kernel: one workgroup iterates on a column subset, its
    // breakpoints are checked against all columns afterwards (solo.hip); what a solo launch leaves behind is
    // staged (Workspace::solo_stage) and becomes the state of the solve only once it is verified
    uint32_t solo_off;     // 1: the rest of this solve runs in the resident form (k_la_persist / k_la_iter)
    uint32_t solo_pending; // 1: a solo launch has run and awaits k_la_verify + k_la_vpublish; 2: a replay (already verified) awaits its commit
    uint32_t solo_nlog;    // breakpoints the solo launch logged
    uint32_t cand_scan;    // 1 once tcand / cand_top carry the candidates of a verified scan (before: |c0| ranks)
    uint32_t solo_replay;  // > 0: the last solo launch failed its check after this many good iterations; the next one repeats exactly those
    uint32_t solo_fails;   // solo launches of this solve that failed a check
    uint32_t solo_started; // early form: set by the solo launch when it begins (the passes over A on the second stream wait for it)
    uint32_t subg_active;  // early form: 1 while the solo launches of this solve run on the subset Gram matrix Gs (until their first commit)
    uint32_t solo_where;   // early form: (XCC_ID << 16 | HW_ID) of the solo workgroup + 1 (0 = none): the passes keep off its shader engine
    uint32_t tie_stall;    // a step-length scan met an off-support column whose candidate is exactly 0 while lambda is where the previous
                           // step left it (tie_band): two columns reached the boundary within rounding, this one attains max|c| and the
                           // reference's strict t > 0 (homotopy-cpu.cpp:143-153) skips it for good.  Which implementation hits that is
                           // rounding luck: the host re-runs such a signal in the reference-order engine (reforder.hip).  Not raised for
                           // the column that has just LEFT the support (it sits on the boundary by rule), nor when an off-support column
                           // dominates by a margin (a path already derailed, e.g. by the first-step sign quirk: every rounding agrees)
    float    lambda0;      // ||A^T y||_inf, the first lambda of the solve (the scale of the tie band, ss_hip_device.h: tie_band)
    uint32_t ro_redo;      // reference-order engine: the direction had to be rebuilt from the true signs — q = A^T A d is swept again (k_ro_check)
    // ---- words other workgroups touch concurrently inside a launch: one 128-B line each
    uint32_t ticket_scan; // arrival counter of k_scansel (reset by the last arriver)
    uint32_t pad1_[31];
    uint32_t ticket_gram; // arrival counter of k_gramupd
    uint32_t pad2_[31];
    double   dot;         // a_idx . a_idx of the column being inserted (k_gramupd hand-off)
    uint32_t sub_reason;  // subset / screened forms: why the signal was not reported (bit mask, resident.h: kReason*); single-writer or atomicOr
    uint32_t pad3_[29];
    uint32_t bar_count;   // grid barrier of k_la_iter: arrivals so far in this solve (monotonic)
    uint32_t pad4_[31];
};
// shader engines of the chip as the early form's passes count them: 8 XCDs x 4
constexpr uint32_t kSeCount = 32;
static_assert(sizeof(DevState) == 640, "DevState layout");

// Hand-off area of the resident lookahead kernel (k_la_persist).  Everything that crosses
// workgroups inside a launch is accessed with L2-bypassing (agent-scope atomic) loads and stores.
struct LaSync {
    uint32_t tick;         // iterations published so far in this solve (carried across launches)
    uint32_t pad_[31];
};
static_assert(sizeof(LaSync) == 128, "LaSync layout");
// The allocation continues with the per-workgroup offer slots smax[2][kLaSlotStride] and
// smin[2][kLaSlotStride] (64-bit, indexed by tick parity and workgroup).  A workgroup posts
// (bits(max |c_i|) << 32 | ~i) and later (bits(min t_i) << 32 | i) in its own slots and reads
// everybody's; it sets a slot back to "empty" two exchanges later.  No read-modify-write atomics:
// 2 x 256 of them per iteration on one cache line cost more than everything else in the iteration.
constexpr uint32_t kLaSlotStride = 512;                       // workers at most
constexpr uint64_t kLaSlotEmpty = ~0ull;                      // not posted yet
constexpr uint64_t kLaSlotNone = 0x7f7fffffffffffffull;       // posted: no step-length candidate
constexpr size_t kLaSyncBytes = 128 + 4 * 512 * 8;
constexpr uint32_t kLaLdsSmall = 96;      // support sizes the resident kernel holds in LDS: first tier ...
constexpr uint32_t kLaLdsLarge = 192;     // ... and the one that takes a whole CU's LDS
// Breakpoint log of a solo launch: a header of 2 * kSoloWidth words — the subset's columns (0xffffffff =
// unused position) and their Gram rows (cache slot; the column itself in full-G mode) — then the entries,
// 8 words + four lists of kSoloListPitch words each:
//   [0] K  [1] flags (bit 0: the entry carries a step-length scan)  [2] round  [3] idx picked
//   [4] bits(lambda)  [5] bits(gamma)  [6] bits(best candidate on the support)  [7] its column
//   gam[], subset position[], bits(x_S)[], bits(d_S)[] in sorted-support order
constexpr uint32_t kSoloLogCap = 128;                         // entries per launch (a C2 path of 64 iterations fits one launch)
constexpr uint32_t kSoloListPitch = kLaLdsSmall;              // solo launches run in the first LDS tier only
constexpr uint32_t kSoloEntryWords = 8 + 4 * kSoloListPitch;
constexpr uint32_t kSoloHeaderWords = 512;
// What a solo launch leaves behind is STAGED: the state of the solve (DevState, lists, inverse, x, d,
// membership flags) is only touched by k_la_vpublish, after the log has been verified.  Staging area (words):
//   [0] K  [1] 1 = pick made, inverse update pending (Gram column missing)  [2] iter  [3] bits(lambda)
//   [4] bits(gamma)  [5] idx  [6] rank  [7] added  [8] tick  [9] exit code  [10] done_round
//   then gam[P], bits(x_S)[P], bits(d_S)[P], gam_new[P + 1] (with [1]), inverse [K][K]
// why a launch of k_la_persist ended (staging word [9] of a solo launch)
enum : uint32_t {
    kPsExitNone = 0,       // still iterating
    kPsExitDone = 1,       // the solve terminated
    kPsExitMiss = 2,       // the entering column has no cached Gram column: lookahead sweep needed
    kPsExitGrow = 3,       // the support outgrew the LDS tier of the launch
    kPsExitWait = 4,       // a bounded wait expired (resident form only)
    kPsExitLogFull = 5,    // solo: log full or replay finished — the next launch goes on
    kPsExitHandOver = 6,   // solo: last step of a path / rare ending — the resident form goes on
    kPsExitNothing = 7,    // solo: the launch could not start (support beyond its tier) — nothing staged
    kPsExitResidue = 8,    // resident form, reference mode: a leaving column keeps a rounding residue — k_la_iter goes on
};
constexpr uint32_t kSoloStageHead = 16;
constexpr uint32_t kSoloStageWords = kSoloStageHead + 4 * kSoloListPitch + 1 + kSoloListPitch * kSoloListPitch;
constexpr uint32_t kSoloChunk = 40;                           // breakpoints verified per pass over the Gram rows (a phase of ~33 entries fits one)
constexpr uint32_t kSoloWidth = 256;                          // columns of a solo launch / of a verify workgroup
constexpr uint32_t kCandPerBlock = 4;                         // entrant candidates kept per 256-column block (cand_top)

// optional per-iteration record of the homotopy path (ss_hip_get_trace)
struct TraceEntry {
    uint32_t idx;
    uint32_t added;
    double   gamma;
    double   c_inf;   // lambda at the START of the iteration (the one the scan used)
};

// Per-signal ("slot") strides of the workspace arrays.  Every kernel of the active-set tail
// takes blockIdx.y as the slot and offsets its pointers by these; a single solve is slot 0
// of a one-slot batch.
struct SlotDims {
    uint32_t n_pad;        // c, q, x, d, insup: one row of n_pad per slot
    uint32_t ldm;          // y, r, p: one row of ldm per slot
    uint32_t m;            // rows of A (entries m..ldm-1 of y, r, p are zero padding)
    uint32_t kcap;         // gam/touched: 2*kcap per slot; inv: 2*kcap*kcap; u1/u2/sgn: kcap
    uint32_t b_pad;        // rows of the r-block (the p-block / q-block start b_pad rows later)
    uint32_t pmax_stride;  // partial (max |c|, index) pairs per slot
    uint32_t pmin_stride;  // partial (min gamma, index) pairs per slot
};

template <typename T>
struct Workspace {
    uint32_t b_cap = 0;    // slots allocated
    SlotDims dims{};
    // per-slot vectors
    T* y = nullptr;        // [b_cap][ldm]       signals, zero padded
    T* rhs = nullptr;      // [2][b_pad][ldm]    r = y - A x (block 0) and p = A d (block 1)
    T* cq = nullptr;       // [2][b_pad][n_pad]  correlations A^T r (block 0) and A^T A d (block 1)
    T* c = nullptr;        // = cq
    T* q = nullptr;        // = cq + b_pad*n_pad
    T* x = nullptr;        // [b_cap][n_pad]     dense solutions
    T* d = nullptr;        // [b_cap][n_pad]     dense directions (non-zero on Gamma only)
    uint8_t* insup = nullptr;   // [b_cap][n_pad] membership flags of Gamma
    // sweep / scan partial reductions
    T* pmax_val = nullptr;       uint32_t* pmax_idx = nullptr;   // [b_cap][pmax_stride]
    T* pmin_val = nullptr;       uint32_t* pmin_idx = nullptr;   // [b_cap][pmin_stride]
    // active set
    uint32_t kcap = 0;
    uint32_t* gam = nullptr;      // [b_cap][2][kcap] sorted support (lambda_indices), ping-pong with inv
    uint32_t* touched = nullptr;  // [b_cap][2][kcap] sorted, every column ever inserted
    T* inv[2] = { nullptr, nullptr };  // [b_cap][2][kcap][kcap] ping-pong (A_S^T A_S)^-1; inv[1] = inv[0] + kcap^2
    T* u1 = nullptr;              // [b_cap][kcap]
    T* u2 = nullptr;              // [b_cap][kcap]
    T* sgn = nullptr;             // [b_cap][kcap]
    DevState* st = nullptr;       // [b_cap]
    uint32_t* ndone = nullptr;    // number of slots that raised `done` in the current (batch) solve
    // lookahead engine (fp32 single-signal): cache of Gram columns g_j = A^T a_j
    T* gcache = nullptr;          // [gcap][n_pad]
    uint32_t gcap = 0;
    uint32_t gpitch = 0;          // row pitch of gcache in elements (n_pad rounded up to 1024)
    int32_t* slot_of = nullptr;   // [n_pad] cache slot of a column, -1 = not cached
    T* c0 = nullptr;              // [n_pad] A^T y
    T* tcand = nullptr;           // [n_pad] per-column step-length candidate of the last scan
    uint32_t* sw_list = nullptr;  // [128] rcols[64] then drows[64] of the next lookahead sweep (32 or 64 columns)
    uint32_t* sw_list2 = nullptr; // [128] the same for the fill-in sweeps of the early form (columns a solo launch used beyond the first 64)
    uint32_t* sub_cols = nullptr; // [kSoloWidth] early form: the columns of the first solo launch (position 0 = the first pick)
    float* subg = nullptr;        // [kSoloWidth][kSoloWidth] early form: Gs = A_S^T A_S of those columns (subgram.hip)
    LaSync* la_sync = nullptr;    // hand-off area of k_la_persist, followed by the offer slots
    T* cq_alt = nullptr;          // [2][n_pad] second (c, q) pair: k_la_persist alternates by tick parity
    // full-G mode of the single-signal engine (fp32): the context's G = A^T A serves as the cache — every
    // column is "cached" in row = its own index, so no lookahead sweep is ever needed.  While a solve runs
    // in this mode gcache / gpitch / slot_of point at G / its pitch / slot_identity; the *_own fields keep
    // the workspace's own cache.
    bool gram_is_full = false;
    T* gcache_own = nullptr;
    uint32_t gpitch_own = 0;
    int32_t* slot_of_own = nullptr;
    int32_t* slot_identity = nullptr;   // [n_pad] 0, 1, 2, ...
    // speculative form (solo.hip)
    uint32_t* slot_col = nullptr; // [gcap] column held by each cache slot
    uint32_t* solo_log = nullptr; // [kSoloHeaderWords] + [kSoloLogCap][kSoloEntryWords]
    uint8_t* sub_pos = nullptr;   // [n_pad] subset position of a column in the last solo launch (valid iff the header agrees)
    uint32_t* solo_stage = nullptr; // [kSoloStageWords] staged hand-over of a solo launch
    uint32_t* v_max = nullptr;    // [kSoloLogCap][nvwg] per-workgroup max |c| (bits) of every logged breakpoint
    uint64_t* v_min = nullptr;    // [kSoloLogCap][nvwg] per-workgroup best step-length candidate (bits << 32 | column)
    uint64_t* cand_top = nullptr; // [nvwg][kCandPerBlock] per-workgroup best entrant candidates (ordered key << 32 | column)
    uint32_t nvwg = 0;            // workgroups of k_la_verify = ceil(n / kSoloWidth)
    uint64_t* la_dbg = nullptr;   // [1024][8] stage timestamps of k_la_iter (option "la_debug"), else null
    uint32_t la_nparts = 0;       // partial maxima written by the last k_la_cq launch
    uint32_t* tile_skip = nullptr; // [b_pad/128 + 1] compact list of GEMM row tiles with a running signal + count
    TraceEntry* trace = nullptr;  // [trace_cap] when tracing is on
    uint32_t trace_cap = 0;
};

struct SweepConfig {
    int variant = 0;
};

}  // namespace sship

struct ss_hip_ctx {
    // batched Gram form: the full G = A^T A ([n_pad][gram_pitch] fp32, made by the first large batch) and
    // the per-signal c0 = A^T y rows of the current chunk
    float* gram_full = nullptr;
    uint32_t gram_pitch = 0;
    // G's memory reserved ahead of its first use: a context that has received a batch of >= 4 signals will likely receive the large
    // one that forms G — the allocation (the driver clears fresh VRAM: ~0.5 s for 17 GiB) then runs on a helper thread beside the
    // batches before it instead of in front of the first large one (option gram_reserve; only where G is a small share of the HBM)
    int screen_rescue = 1;                   // a declined screened solve (positions ran out / an outside column beat a state) is scanned for the columns the
                                             // ranking missed and repeated once with those in the subset (screen.hip: launch_screen_rescue_scan)
    int screen_first8 = 1;                   // the screened form's ranking pass over an fp8 copy of A (screen.hip: k_scr_first8); 0 = over the fp16 copy
    int first_pass_elem_bytes = 2;           // what the last reduced-precision first pass read per entry of A (statistics)
    void* gram_reserve_thread = nullptr;     // std::thread*
    float* gram_reserved = nullptr;          // what that thread obtained (read after join)
    int gram_reserve = 1;
    float* c0_batch = nullptr;
    void* sub_buf = nullptr;          // subset form (subbatch.hip): subsets, first picks, breakpoint logs of a chunk
    size_t sub_buf_bytes = 0;
    int sub_attr_set = -1;
    // the subset form steps aside where it does not pay: after a chunk (or a run of single solves) of which it had to hand
    // back more than a third, the next 8 chunks (64 solves) go the other way at once, then it is tried again
    uint32_t sub_off_chunks = 0, sub_off_solves = 0, sub_seen = 0, sub_failed = 0;
    uint32_t res_off_solves = 0, res_seen = 0, res_failed = 0;   // ... the same for the resident tier of the fp64 screened form
    int screen_recheck = 1;           // option: 1 = columns the screened form's half-precision certificate cannot vouch for are re-checked exactly in fp32
                                      // (screen.hip: k_scr_recheck) instead of failing the signal
    int screen_resident = 1;          // option: 1 = the screened forms run their path in the resident kernel (resident.hip), 0 = the forms before it
    hipEvent_t ev_sub_sel = nullptr;  // profiling: between the subset form's selection and its solves
    hipEvent_t ev_c0a = nullptr, ev_c0b = nullptr;   // profiling: around the batch GEMM c0 = A^T y of a chunk
    int batch_subset = 1;             // option: 1 = large Gram-form batches run in the subset form (one workgroup per signal + a check over all columns)
    void* screen = nullptr;           // sship::ScreenState* (screen.hip): fp16 copy of A, column norms, the subset's Gram matrix, residual block
    int screen_first16 = 1;           // option: the screened form's first pass (A^T y over all columns) reads the half-precision copy too
    int screen_single = 1;            // option: 1 = single fp32 signals on large dictionaries take the screened form (screen.hip), 2 = on every shape
                                      // the form can run on (tests), 0 = never
    int screen_failed_alloc = 0;      // the preparation did not fit: not tried again
    uint64_t batch_signals_seen = 0;  // signals this context has received in batches (when G starts to pay: solve_batch_dispatch)
    void* sub_dbg = nullptr;          // developer aid (SS_HIP_SUB_STAMPS): 16 x u64 cycle stamps of k_sub_solve's phases (slot 0)
    int batch_screen = 1;             // option: 1 = fp32 batches of 4 .. batch_gram_min - 1 signals (no G) run in the screened form, 64 per chunk
    // state log of the launch-per-iteration form (k_la_iter; on in the sub-context of the fp64 screened form, screen.hip):
    // cnt[cap] u32, lambda[cap] f64, lambda_prev - gamma_prev [cap] f64, cols[cap][kmax] u32, vals[cap][kmax] T — the state every launch starts from
    void* slog = nullptr;
    uint32_t slog_cap = 0, slog_kmax = 0;
    // narrow fp64 dictionaries (the same sub-context): the passes split their rows over pass_ksplit workgroups per column tile,
    // partial sums in pass_part ([pass_ksplit][64][n_pad] doubles), added up in order (gemm.hip: launch_gemm_split_f64)
    int pass_ksplit = 0;
    void* pass_part = nullptr;
    size_t c0_batch_rows = 0;
    // column form of mid-size batches: cache of Gram columns, row tables, pass lists (grown on demand)
    float* bcol_cache = nullptr;
    size_t bcol_cache_rows = 0;
    int32_t* bcol_slot = nullptr;
    size_t bcol_slot_rows = 0;
    uint32_t* bcol_lists = nullptr;   // rcols[1024] then drows[1024]
    int early_pass = 2;               // option: tiling of the early form's passes (2 = 128-column LDS tiles, 3 workgroups per CU; 0 = k_gemm32e)
    int early_se = 1;                 // option: the early form's passes are dealt out by shader engine around the solo workgroup
    hipStream_t stream3 = nullptr;    // ... third stream (the tiles of the other shader engines), its events, the per-SE counters
    hipStream_t stream4 = nullptr;    // ... and a fourth for the two left-over tiles of each pass (VALU)
    hipEvent_t ev_join4 = nullptr;
    hipEvent_t ev_gate = nullptr, ev_b0 = nullptr, ev_join3 = nullptr;
    uint32_t* se_count = nullptr;     // [2][kSeCount + 2] device counters, one set per pass: arrivals per SE, arrivals in all, tiles taken
    int cq_rows = 2;                  // option: rows of G a thread of the fused batched Gram-form pass has in flight (1, 2, 3, 4, 8)
    int cq_cols = 16;                 // option: columns per thread of that pass (4, 8, 16, 32): a workgroup reads runs of 256 * cq_cols columns of a
                                      // row of G — 16 KiB at 16 (measured 9500 signals/s at 8192 x 65536 x 4096 against 7560 with 4: DESIGN.md §3.6)
    int cq_vec4 = 0;                  // option (A/B): threads of the batched Gram-form pass own four consecutive columns (16-byte loads) instead of four strided ones
    int batch_fused_scan = 1;         // option: the batched Gram forms scan inside the Gram-form pass (k_la_cqs: c, q stay in registers)
    int scan_blocks = 8;              // option: workgroups per slot of the batched Gram form's scan (0 = one per 1024 columns)
    int sweep_f64_variant = 0;        // option: tiling of the 32-column fp64 pass (0 = 256 columns / 512 threads / 1 per CU; 1, 2 = 128 / 256 / 2, 3 per CU)
    uint64_t solo_seen = 0, solo_failed = 0;   // speculative solves / failed checks since the form was last switched off (private: not the statistics)
    int early_adapt = 1;              // option: the early form's second pass takes its columns from the solo launch's progress (0 = from |c0|)
    int batch_cols_min = 24;          // option: smallest fp32 batch that runs in lock-step in the column form (0 = never)
    int batch_cols_max = 0;           // largest one (0 = no limit: larger batches run in chunks of <= 448 signals); batches of
                                      // batch_gram_min signals or more form G instead when that is allowed
    int bcol_chunk = 448;             // signals per chunk of the batch being dispatched (set by the dispatcher)
    unsigned char* rec_stage = nullptr;   // compact output: device staging of the records of one chunk
    size_t rec_stage_bytes = 0;
    long gram_full_gib = 64;     // option: largest G the batched Gram form may allocate
    int batch_gram_min = 512;    // option: smallest batch that pays for making G (0 = never)
    long gram_full_after = 0;    // option: single-signal solves on this context after which G is made for them too (0 = never: opt-in — 17 GiB and 0.3-0.55 s at C2)
    int gram_single = 1;         // option: 1 = single-signal solves use G as their Gram-column cache once it exists, 0 = never
    int gram_symmetric = 1;      // option: 1 = G is formed from the tiles on and above the diagonal + mirrored store, 0 = full product
    uint64_t single_solves = 0;  // single-signal solves since create (the trigger of gram_full_after; not a statistic)
    int colshard_fail_prepare = 0;   // test option: this rank's column-sharded solves fail in their preparation (the ranks must all leave)
    void* colshard = nullptr;    // sship::ColShard* of a column-sharded context (colshard.hip): shard description, communicator, replicated active set
    int kind = 0;            // 0 = Homotopy / OMP context, 1 = IRLS context
    void* irls = nullptr;    // sship::IrlsState<T>* of an IRLS context
    int device = 0;
    int is_f64 = 0;
    size_t m = 0, n = 0;
    uint32_t ldm = 0;        // column pitch in elements (multiple of kRowPad)
    uint32_t n_pad = 0;      // padded column count (multiple of kColPad)
    void* At = nullptr;      // [n_pad][ldm] column-contiguous device copy of A
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;           // early form: the passes over A that run beside the solo launch
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int num_cus = 256;
    size_t lds_per_block = 65536;

    // options
    int sweep_variant = 5;   // 16 waves x 4 columns, 2-stage ring, 1 workgroup per CU: fastest on MI355X (profiles/)
    long temporal_cols = 0;  // leading dictionary columns swept with cache-allocating loads (rest: nt)
    int lookahead = 4;
    int strict_sign = 0;
    int zero_on_removal = 0; // 0 = the reference's x + gamma*d residue on a leaving column (homotopy-cpu.cpp:252); 1 = exact 0 (opt-in)
    int tie_guard = 0;       // 0 = the reference's strict t > 0 (homotopy-cpu.cpp:135,145,151); 1 = zero-length step on an exact tie (opt-in)
    int ro_staged = 1;        // option: 1 = the reference-order sweep stages the dictionary through LDS (coalesced loads), 0 = direct 16-byte loads
    int ro_slots = 8;         // option: signals the reference-order engine runs in lock-step per pass over A (1..8 in fp32, ..4 in fp64)
    int ro_force_resweep = 0; // developer option: the reference-order engine treats every sign check as failed (the second sweep of an iteration always runs)
    int tie_rerun = 1;       // 1 = a solve whose scan met a tie stall (DevState::tie_stall) is re-run in the reference-order engine
    int profiling = 0;
    int profile_every = 1;   // with profiling on, time every k-th fused sweep
    int profile_solve_every = 1;   // ... and only every k-th solve at all
    uint64_t prof_solve_tick = 0;
    long cache_mib = 2048;   // budget of the lookahead engine's Gram-column cache
    int engine = 1;          // fp32 single-signal Homotopy: 1 = lookahead (cached Gram columns) unless the tolerance is too tight for it, 2 = lookahead always, 0 = one fused sweep per iteration
    int early_solo = 1;        // option: 1 = early form of the speculative engine (iterations on the subset Gram matrix beside the passes over A)
    int sweep_cols_f64 = 64;   // option: columns per lookahead sweep in double precision (64: the pass is MFMA-bound either way; 32)
    int sweep_cols_f64_late = 32;     // option: ... of the third and later passes of a solve (late misses are sparse)
    int early_probe = 0;       // developer aid (option): 1 = early form without overlap (passes first, then the solo launch)
    int first_sweep_cols = 32; // option: columns of the first lookahead sweep of a fp32 solve (64: one MFMA-bound pass instead of two HBM-bound ones; 32)
    int sweep32_variant = 0; // lookahead sweep tiling: 0 = 256 columns x 512 threads (1 per CU), 1 / 2 = 128 columns x 256 threads (2 / 3 per CU)
    int la_fused = 3;        // lookahead engine: 3 = speculative form of the resident kernel (one workgroup + verification of every breakpoint, solo.hip), 2 = resident kernel (k_la_persist), 1 = one kernel per iteration (k_la_iter), 0 = scan / update / cq kernels
    int solo_subset = 256;   // option (tests): columns a solo launch may hold (<= 256; small values provoke verification failures)
    int solo_full_gram = 0;  // option (tests): let the speculative form run with the full Gram matrix as the cache too
    int solo_off_solves = 0; // solves left for which the speculative form stays off after repeated verification failures
    int batch_min = 192;     // batches of at least this many fp32 signals run in lock-step on the MFMA GEMM (below: one lookahead solve per signal, ~2.4 ms each at C2, is faster)
    int batch_chunk = 4096;  // signals processed together by the batched path
    int tracing = 0;
    std::vector<sship::TraceEntry> last_trace;   // host copy of the last solve's path

    // workspace (type-erased; Workspace<float> or Workspace<double>)
    void* ws = nullptr;
    // pinned, device-mapped: [0] = last round the device started (fused lookahead engine: iteration
    // launches executed), [1] = done, [2] = iterations waiting for a lookahead sweep so far.
    // Written by the device with system-scope stores, polled by the host loop (no copies in the stream).
    uint32_t* host_flags = nullptr;
    void* hs_pinned = nullptr;        // pinned landing place of the end-of-solve DevState copy (sizeof(DevState))
    void* hs_mapped = nullptr;        // its device address (k_epilogue writes the state there)
    uint32_t* dev_flags = nullptr;    // device address of host_flags
    std::vector<hipEvent_t> prof_events;   // pairs (start, stop) for sweeps of the current solve
    std::vector<int> prof_kind;            // 2 = fused sweep, 1 = single-RHS sweep
    hipEvent_t ev_solve0 = nullptr, ev_solve1 = nullptr;
    int solo_attr_set = -1;                // dynamic-LDS attribute of the solo kernel: -1 not tried, 0 refused, 1 set
    int persist_workers[2] = { -1, -1 };   // worker workgroups of k_la_persist per LDS tier (-1 = not queried, 0 = unusable)

    ss_hip_stats stats{};
};

namespace sship {

// ---- launchers implemented in sweep.hip ----------------------------------------
// [out0, out1] = A^T [rhs0, rhs1]; out1 == nullptr selects the single-RHS kernel.
// pmax_* receive one (max |out0|, first index) pair per workgroup; *nblocks_out is the
// number of pairs written.  st may be nullptr (standalone sweep).
template <typename T>
hipError_t launch_sweep(const ss_hip_ctx* ctx, const T* rhs, size_t rhs_stride, int nrhs, T* out0, T* out1,
                        T* pmax_val, uint32_t* pmax_idx, uint32_t* nblocks_out,
                        const DevState* st);
size_t sweep_max_lds_bytes();

// ---- launchers implemented in activeset.hip ------------------------------------
// nslots = signals in flight (grid.y); nparts = partial maxima per slot in pmax_*
template <typename T>
hipError_t launch_init(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t nparts, T tol);
template <typename T>
hipError_t launch_rp(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots);
template <typename T>
hipError_t launch_iteration_tail(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t round,
                                 uint32_t nparts, T tol, uint32_t max_iter);
// the scan + select + toggle + x update of one iteration on its own (reference-order engine)
template <typename T>
hipError_t launch_scansel_plain(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t round, uint32_t nparts, T tol, uint32_t max_iter);
// ---- reference-order engine (reforder.hip): every reduction in the documented 8-partial order ------------
// reference-order engine (reforder.hip), for the first nslots slots of the workspace layout.
// launch_ro_sweep: block b of slot s is v + b * blk_stride + s * ldm -> out + b * out_blk + s * n_pad (nblk = 1 or 2 blocks);
// gate: only slots whose DevState::ro_redo is raised
template <typename T>
hipError_t launch_ro_sweep(const ss_hip_ctx* ctx, const T* v, size_t blk_stride, T* out, size_t out_blk, uint32_t n_pad, int nblk,
                           uint32_t nslots, T* pmax_val, uint32_t* pmax_idx, uint32_t pmax_stride, uint32_t* nblocks_out,
                           const DevState* st, bool gate);
template <typename T> hipError_t launch_ro_mv(const ss_hip_ctx* ctx, Workspace<T>& ws, int mode, uint32_t nslots, bool gate);   // 0: r = y - A x, 1: p = A d
template <typename T> hipError_t launch_ro_init(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t nparts, T tol);
// inverse update, then the signs taken from c - gamma q and the direction built from them
template <typename T> hipError_t launch_ro_update(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t round, T tol);
// lambda and the while-test from the correlations just re-computed; the signs the direction was built from are checked
// against them (a mismatch rebuilds the direction and raises DevState::ro_redo)
template <typename T> hipError_t launch_ro_check(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t nparts, T tol, uint32_t max_iter);
// one whole round: r, p, the fused sweep, the check (+ the gated second sweep), scan + toggle, inverse + direction
template <typename T>
hipError_t launch_ro_round(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t round, uint32_t nparts, T tol, uint32_t max_iter);
uint32_t ro_slots_max(const ss_hip_ctx* ctx, bool f64);
// subset form of the batched Gram form (subbatch.hip): select + solve + verify for the first nslots slots; c0 = A^T y of every slot
bool sub_form_usable(ss_hip_ctx* ctx);
size_t sub_buffer_bytes(uint32_t nslots);
// the form's buffers inside ss_hip_ctx::sub_buf (sized by sub_buffer_bytes(nslots)) and its three stages on the context's stream
struct SubBufs { uint32_t* sub; uint32_t* fpick; float* fval; uint32_t* hdr; uint32_t* pcol; float* LX; float* LD; };
SubBufs sub_bufs(ss_hip_ctx* ctx, uint32_t nslots);
hipError_t launch_sub_select(ss_hip_ctx* ctx, const SubBufs& B, uint32_t nslots, const float* c0, float* thr_out = nullptr, const float* wmax = nullptr,
                             uint32_t nwmax = 0);
hipError_t launch_sub_solve(ss_hip_ctx* ctx, Workspace<float>& ws, const SubBufs& B, uint32_t nslots, const float* G, uint32_t gpitch, int gsub,
                            const float* c0, float tol, uint32_t max_iter, uint32_t g_slot_stride = 0);
hipError_t launch_sub_finish(ss_hip_ctx* ctx, Workspace<float>& ws, uint32_t nslots);
hipError_t launch_sub_finish_st(ss_hip_ctx* ctx, DevState* st, uint32_t nslots);
hipError_t launch_select_top(ss_hip_ctx* ctx, const float* v, uint32_t n, uint32_t n_pad, uint32_t nsel, uint32_t* sub, uint32_t* fpick, float* fval,
                             float* thr_out = nullptr, const float* wmax = nullptr, uint32_t nwmax = 0);
// screened form of ONE signal (screen.hip): the subset form on the subset's own Gram matrix (formed from A), every state of
// the path then screened against all columns by one pass over a half-precision copy of A with a rigorous error bound
bool screen_form_usable(ss_hip_ctx* ctx);                 // shape / option test + one-time preparation (fp16 copy of A, column norms)
bool screen_first16_usable(const ss_hip_ctx* ctx);        // ... with the FIRST pass (A^T y) over the half-precision copy too
hipError_t launch_screen_form(ss_hip_ctx* ctx, Workspace<float>& ws, float tol, uint32_t max_iter, bool first16, bool finish, hipEvent_t e0 = nullptr,
                              hipEvent_t e1 = nullptr, hipEvent_t e2 = nullptr, hipEvent_t e3 = nullptr, hipEvent_t e4 = nullptr, hipEvent_t e5 = nullptr,
                              bool omp = false, bool rescue = false);
// the rescue of a declined solve: the scan of its log for the columns the ranking missed (-> their number), see screen.hip
hipError_t launch_screen_rescue_scan(ss_hip_ctx* ctx, Workspace<float>& ws, float tol, bool from_recheck, uint32_t* count_out);
uint32_t screen_rescue_cap();
// a batch chunk of nslots <= screen_batch_cap() signals in the screened form: c0 = A^T y of every slot in c0_all ([nslots][n_pad]), the
// signals in ws.y; the slots' verdicts in their states (k_sub_finish) like the subset form's
uint32_t screen_batch_cap();
hipError_t launch_screen_batch(ss_hip_ctx* ctx, Workspace<float>& ws, uint32_t nslots, const float* c0_all, float tol, uint32_t max_iter);
void screen_free(ss_hip_ctx* ctx);
// fp64: the path is solved by the fp64 engine on a sub-dictionary (a context of its own: the kS64Sub columns with the largest
// |A^T y|), its states are logged (ss_hip_ctx::slog) and certified against all columns by the same fp16 pass
bool screen64_usable(ss_hip_ctx* ctx);
ss_hip_ctx* screen64_sub(ss_hip_ctx* ctx);
double* screen64_xsub(ss_hip_ctx* ctx);
hipError_t screen64_gather(ss_hip_ctx* ctx, const double* c0, const double* y, hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr);   // (c0 = nullptr: the first pass over the fp16 copy, here; e0, e1 around it)
hipError_t screen64_certify(ss_hip_ctx* ctx, Workspace<double>& ws, const double* y, uint32_t T, double tol, double c_inf, uint32_t K,
                            hipEvent_t e2 = nullptr, hipEvent_t e3 = nullptr, bool omp = false, bool first16 = false);
// fp64, resident tier (resident.hip): the path on the 256 best-ranked columns in ONE workgroup, everything queued in one go
bool screen64_resident_usable(ss_hip_ctx* ctx);
hipError_t launch_screen64_resident(ss_hip_ctx* ctx, Workspace<double>& ws, double tol, uint32_t max_iter, bool first16, bool omp,
                                    hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, hipEvent_t e2 = nullptr, hipEvent_t e3 = nullptr,
                                    hipEvent_t e4 = nullptr, hipEvent_t e5 = nullptr, bool rescue = false);
hipError_t launch_screen64_rescue_scan(ss_hip_ctx* ctx, Workspace<double>& ws, double tol, bool from_recheck, uint32_t* count_out);
// fp64 batches in the resident tier: a chunk of nslots <= screen64_batch_cap() signals (in ws.y) — one pass over the fp16 copy ranks every signal's
// columns, the chunk's paths run side by side, each signal's states are certified by a screening pass of its own; verdicts in the slots' states
bool screen64_batch_usable(ss_hip_ctx* ctx);
uint32_t screen64_batch_cap();
hipError_t launch_screen64_batch(ss_hip_ctx* ctx, Workspace<double>& ws, uint32_t nslots, double tol, uint32_t max_iter);
void screen_debug_recheck(ss_hip_ctx* ctx);               // developer aid (SS_HIP_SUB_DEBUG)
double screen_read_headroom(ss_hip_ctx* ctx);             // largest (|c~| + eps) / bound of the last screened solve (synchronises)
hipError_t launch_sub_form(ss_hip_ctx* ctx, Workspace<float>& ws, uint32_t nslots, const float* c0, float tol, uint32_t max_iter,
                           hipEvent_t e0 = nullptr, hipEvent_t e1 = nullptr, hipEvent_t e2 = nullptr);      // signals one pass of the engine can carry (1 without the LDS-staged sweep)
// list[0..count) = 128-row tiles that still hold a running signal, list[rows/128] = count
hipError_t launch_tile_list(const ss_hip_ctx* ctx, const DevState* st, uint32_t nslots, uint32_t rows,
                            uint32_t* list);
template <typename T>
hipError_t launch_omp_tail(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t round,
                           uint32_t nparts, T tol, uint32_t max_iter);
// lookahead engine launchers (activeset.hip); see homotopy.hip for the round structure
template <typename T>
hipError_t launch_la_init_pick(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nparts, T tol, bool full_gram = false);
template <typename T>
hipError_t launch_la_top(const ss_hip_ctx* ctx, Workspace<T>& ws, int init_mode, uint32_t nsel = 32);
template <typename T>
hipError_t launch_la_update(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t round, T tol);
// one launch for everything a lookahead solve clears before its first sweep (and r = y)
template <typename T>
hipError_t launch_la_reset(const ss_hip_ctx* ctx, Workspace<T>& ws, bool clear_slots, const T* y_user = nullptr, ptrdiff_t incy = 1);   // y_user: the caller's signal, if it is on the device (else ws.y holds it)
template <typename T>
hipError_t launch_la_cq(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t* nparts_out);
// fused iteration of the lookahead engine (c, q from the cache; scan; select; update)
template <typename T>
hipError_t launch_la_iter(const ss_hip_ctx* ctx, Workspace<T>& ws, T tol, uint32_t max_iter);
template <typename T>
hipError_t launch_la_scansel(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t round, uint32_t nparts, T tol,
                             uint32_t max_iter);
// batched Gram form (activeset.hip): c, q of every slot from rows of the full G = A^T A; scan + inverse update
template <typename T>
hipError_t launch_gram_guard_batched(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, T tol);
// column form of a mid-size batch (no G): the Gram columns of the entering columns are formed round by round,
// 64 per pass over A, into rows of a cache the batch owns (row = round * nslots + slot, pitch n_pad)
struct BatchCols {
    float* cache = nullptr;        // [rounds][nslots][pitch]
    uint32_t pitch = 0;            // row pitch in floats: n_pad rounded up to 1024 (k_la_cq reads whole 1024-column chunks of a row)
    int32_t* bslot = nullptr;      // [nslots][n_pad]: column -> cache row of that slot (-1 = not cached)
    uint32_t* rcols = nullptr;     // [cap] entering columns of the round, compacted (0xffffffff = none)
    uint32_t* drows = nullptr;     // [cap] their cache rows
    uint32_t cap = 0;              // nslots rounded up to 64
    uint32_t row_base = 0;         // round * nslots
};
hipError_t launch_batch_cols(const ss_hip_ctx* ctx, const DevState* st, uint32_t nslots, uint32_t row_base, bool first,
                             int32_t* bslot, uint32_t* rcols, uint32_t* drows, uint32_t cap);
hipError_t launch_batch_passes(const ss_hip_ctx* ctx, const BatchCols* cols, uint32_t nslots);
// round != 0: the fused form (k_la_cqs: Gram-form pass + lambda + scan + pick of that round in one launch; the tail
// is then called with scan_done = true)
bool cqs_usable(const ss_hip_ctx* ctx, uint32_t pmin_stride);
template <typename T>
hipError_t launch_cq_gram_batched(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, const T* G, uint32_t gpitch,
                                  const T* c0b, uint32_t* nparts_out, const int32_t* bslot = nullptr,
                                  uint32_t round = 0, T tol = T(0), uint32_t max_iter = 0);
template <typename T>
hipError_t launch_tail_gram_batched(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t round, uint32_t nparts,
                                    T tol, uint32_t max_iter, const T* G, uint32_t gpitch, const BatchCols* cols = nullptr,
                                    bool scan_done = false);
// orthogonal matching pursuit in Gram form: one launch per iteration, and the pending update after a fetch
template <typename T>
hipError_t launch_la_omp(const ss_hip_ctx* ctx, Workspace<T>& ws, T tol, uint32_t max_iter);
template <typename T>
hipError_t launch_la_omp_update(const ss_hip_ctx* ctx, Workspace<T>& ws, T tol);
// resident form (persist.hip): one launch runs iterations until the solve ends, a Gram column is
// missing or the support outgrows `lds_cols`; returns hipErrorInvalidConfiguration if the device
// cannot keep the whole grid resident for this n
hipError_t launch_la_persist_f32(ss_hip_ctx* ctx, Workspace<float>& ws, float tol, uint32_t max_iter, uint32_t lds_cols, bool after_solo = false);
// early form (solo.hip / subgram.hip): the subset of the first solo launch + the slots of the first 64 Gram columns;
// the gate of the second stream; the slots of the columns the launch used beyond those 64
hipError_t launch_subset_pick_f32(ss_hip_ctx* ctx, Workspace<float>& ws);
hipError_t launch_wait_started(ss_hip_ctx* ctx, Workspace<float>& ws, hipStream_t on);
hipError_t launch_missing_cols_f32(ss_hip_ctx* ctx, Workspace<float>& ws);
// the barrier-free 32-column pass (one 32-column tile per single-wave workgroup) on a given stream, ungated
hipError_t set_pass_debug(uint64_t* buf);      // developer aid: per-workgroup trace of the fp32 lookahead passes (gemm.hip)
hipError_t launch_pick_pass_b_f32(ss_hip_ctx* ctx, Workspace<float>& ws, hipStream_t on);
hipError_t launch_gemm32w_on(const ss_hip_ctx* ctx, hipStream_t on, const uint32_t* rcols, const uint32_t* drows, float* D, uint32_t ldd,
                             uint32_t tiles128 = 0);
inline uint32_t early_se_wgs(const ss_hip_ctx* ctx) { return ctx->early_se == 2 ? kSeCount : 2u * kSeCount; }   // (3: as 1, any tile count)
// tiles first .. ntiles-1 of the same pass, two per shader engine except the solo workgroup's (early form)
hipError_t launch_gemm32se_on(const ss_hip_ctx* ctx, hipStream_t on, const uint32_t* rcols, const uint32_t* drows, float* D, uint32_t ldd,
                              uint32_t first, uint32_t ntiles, const DevState* st, uint32_t* se_count, uint32_t quota = 2u);
// tiles [first, last) of a pass as a plain launch (the partial round of a wide dictionary)
hipError_t launch_gemm32range_on(const ss_hip_ctx* ctx, hipStream_t on, const uint32_t* rcols, const uint32_t* drows, float* D, uint32_t ldd,
                                 uint32_t first, uint32_t last);
hipError_t launch_wait_count(hipStream_t on, const uint32_t* counter, uint32_t target, const DevState* st);
hipError_t launch_cols_gram_on(const ss_hip_ctx* ctx, hipStream_t on, uint32_t c0, uint32_t ncols, const uint32_t* rcols,
                               const uint32_t* drows, float* D, uint32_t ldd);
// speculative form: k_la_persist<solo> (one workgroup on a column subset), then k_la_verify and
// k_la_vpublish (solo.hip), which check the logged breakpoints against all columns and release or
// revoke the outcome; launch_la_cand_init seeds the subset ranking from |c0|
hipError_t launch_la_solo_f32(ss_hip_ctx* ctx, Workspace<float>& ws, float tol, uint32_t max_iter);
hipError_t launch_la_verify_f32(ss_hip_ctx* ctx, Workspace<float>& ws);
hipError_t launch_la_cand_init_f32(ss_hip_ctx* ctx, Workspace<float>& ws);
// the columns of the next lookahead sweep from the per-block candidate tops (instead of k_la_top's scan)
hipError_t launch_la_top_cand_f32(ss_hip_ctx* ctx, Workspace<float>& ws, uint32_t nsel = 32);
bool la_solo_usable(ss_hip_ctx* ctx);
// true if k_la_persist can serve this context (column count vs resident workgroups)
bool la_persist_usable(ss_hip_ctx* ctx, uint32_t lds_cols);
// per-slot partial (max |c|, first index) over chunks of the correlation rows (batched path)
template <typename T>
hipError_t launch_absmax(const ss_hip_ctx* ctx, Workspace<T>& ws, uint32_t nslots, uint32_t* nparts_out);
template <typename T>
hipError_t launch_gemv_n(const ss_hip_ctx* ctx, const T* x_dev, T* y_dev);

// ---- launcher implemented in gemm.hip --------------------------------------------
// D[Mg][ldd] = R[Mg][ldr] * At^T on the MFMA units (fp32).  Mg % 128 == 0.  tile_list (may be
// null = all tiles): compact list of the row tiles to compute, count at tile_list[Mg/128].
hipError_t launch_gemm_tn_f32(const ss_hip_ctx* ctx, const float* R, uint32_t Mg, uint32_t ldr,
                              float* D, uint32_t ldd, const uint32_t* tile_list, bool blocked = true);

// G[n_pad][ldd] = At · At^T (the full Gram matrix) from the tiles on and above the diagonal + mirrored stores
hipError_t launch_gemm_sym_f32(const ss_hip_ctx* ctx, float* G, uint32_t ldd);

// D[drows[s]][:] = At · At[rcols[s]][:] for s < 32: 32 right-hand sides in one HBM-bound pass
// (rcols / drows live on the device; 0xffffffff entries are skipped; with st != null the launch
// is a no-op unless st->need_sweep is set and the solve is still running)
hipError_t launch_gemm32_tn_f32(const ss_hip_ctx* ctx, const uint32_t* rcols, const uint32_t* drows,
                                float* D, uint32_t ldd, const DevState* st);

// the same pass with 64 right-hand sides (rcols / drows hold 64 entries): the first lookahead sweep of a solve
hipError_t launch_gemm64_tn_f32(const ss_hip_ctx* ctx, const uint32_t* rcols, const uint32_t* drows,
                                float* D, uint32_t ldd, const DevState* st);

// ---- subgram.hip: Gs[256][256] = A_S^T A_S of a 256-column subset, bit for bit the lookahead sweep's values
hipError_t launch_subset_gram_f32(const ss_hip_ctx* ctx, const uint32_t* cols_dev, float* Gs, const DevState* st,
                                  float* seed_base, const int32_t* slot_of, uint32_t gpitch);

// ---- IRLS (irls.hip) ---------------------------------------------------------------------
struct IrlsResult {
    uint32_t iter;
    uint32_t spd_failure;
    double solution_error;
};
// Householder QR of the device copy of A (in place), thin Q, R and Q^T Q; M >= N
template <typename T> hipError_t irls_factor(ss_hip_ctx* ctx);
// the solve: y in irls_y_buffer, x left in irls_x_buffer, report copied to res_host (stream-ordered)
template <typename T> hipError_t irls_solve(ss_hip_ctx* ctx, T tol, uint32_t max_iter, IrlsResult* res_host);
template <typename T> T* irls_y_buffer(ss_hip_ctx* ctx);
template <typename T> T* irls_x_buffer(ss_hip_ctx* ctx);
void irls_free(ss_hip_ctx* ctx);

// the same pass in fp64 (v_mfma_f64_16x16x4_f64), 32 or 64 right-hand sides
hipError_t launch_gemm32_tn_f64(const ss_hip_ctx* ctx, const uint32_t* rcols, const uint32_t* drows,
                                double* D, uint32_t ldd, const DevState* st);
hipError_t launch_gemm64_tn_f64(const ss_hip_ctx* ctx, const uint32_t* rcols, const uint32_t* drows,
                                double* D, uint32_t ldd, const DevState* st);

// ---- helpers implemented in homotopy.hip ---------------------------------------
void set_err(char* err, size_t errlen, const std::string& msg);
// the one-slot fp32 workspace of a context with an active-set capacity of at least kcap (0 / ss_hip_status)
int colshard_workspace(ss_hip_ctx* ctx, uint32_t kcap);
// ---- colshard.hip: releases the column-sharded state of a context (communicator, replicated active set)
void colshard_destroy(ss_hip_ctx* ctx);

}  // namespace sship
