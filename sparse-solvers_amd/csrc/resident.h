// resident.h — declarations of the resident subset solve (resident.hip), shared with screen.hip / homotopy.hip.
#pragma once

#include "ss_hip_internal.h"

namespace sship {

// shape of the resident kernel per element type: subset columns, threads (eight lanes per group of CT columns), positions
// (support columns a path may take), states it may log
template <typename T> struct ResCfg;
template <> struct ResCfg<double> { static constexpr int S = 256, THREADS = 512, CT = 4, PCAP = 136, LOGCAP = 160; };
template <> struct ResCfg<float>  { static constexpr int S = 448, THREADS = 448, CT = 8, PCAP = 72, LOGCAP = 80; };
static_assert(ResCfg<float>::S == (int)kSbS && ResCfg<float>::PCAP == (int)kSbRows && ResCfg<float>::LOGCAP == (int)kSbLog,
              "the fp32 resident kernel writes the subset form's log (subbatch.hip, screen.hip read it)");

constexpr uint32_t kScrFlCap = 1024;              // columns the screened forms' half-precision certificate may leave to the exact re-check (screen.hip)
constexpr uint32_t kSg64MaxSplit = 64;            // row chunks of the fp64 subset Gram matrix at most

// why a subset solve was not reported (DevState::sub_reason, a bit mask; ss_hip_stats counts them)
enum : uint32_t {
    kReasonRemoval = 1u << 0,       // a column would leave the support (the certified forms take regular paths only)
    kReasonPositions = 1u << 1,     // more support columns than the kernel holds (or than the caller's capacity)
    kReasonLog = 1u << 2,           // more states than the log holds
    kReasonGuard = 1u << 3,         // tolerance below the Gram-form guard
    kReasonNoCand = 1u << 4,        // no positive step-length candidate / nothing to pick
    kReasonFirstState = 1u << 5,    // state 0 not certified: columns left out of the subset may reach lambda_0 (crowded first state)
    kReasonIrregular = 1u << 6,     // lambda went up along the path (derailed by the first-step sign quirk) or a removal was logged
    kReasonOverflow = 1u << 7,      // a residual overflowed the half-precision range
    kReasonColumn = 1u << 8,        // the screening pass could not certify a (column, state)
    kReasonTie = 1u << 9,           // the subset's scan met an exact tie
    kReasonRechecked = 1u << 10,    // (not a failure) the exact re-check of flagged columns ran for this signal
    kReasonRepaired = 1u << 11,     // (not a failure) ... and the last step was taken again with a column outside the subset stopping it
};

// the log of a resident solve (device pointers; per slot: hdr [LOGCAP][8] u32, H [LOGCAP][2] = {lambda, gamma} in T (may be null),
// pcol [PCAP] u32, X [LOGCAP][PCAP] = x by position of every state, D (may be null) = the direction by position likewise)
template <typename T> struct ResLog { uint32_t* hdr; T* H; uint32_t* pcol; T* X; T* D; };

template <typename T> bool res_solve_usable();
template <typename T>
hipError_t launch_res_solve(ss_hip_ctx* ctx, uint32_t nslots, const T* Gs, uint32_t gpitch, size_t g_slot_stride, const T* c0, uint32_t c0_stride,
                            const uint32_t* sub, T tol, uint32_t max_iter, uint32_t kcap, const ResLog<T>& log, T* x, uint32_t x_stride,
                            uint32_t* gam2, uint32_t* touched2, DevState* st, TraceEntry* trace, uint32_t trace_cap, bool omp);
// fp64: Gs = A_S^T A_S of the 256 subset columns (part: [kSg64MaxSplit][256][256] scratch) and their exact c0 = a_j . y into the dense c0
// (nslots > 1: a batch — sub, y, gs, c0 and the partials carry a slot dimension: S, ldm, S * S, c0_stride, nsplit * S * S elements per slot)
hipError_t launch_sgram64(ss_hip_ctx* ctx, const uint32_t* sub, const double* y, double* part, double* gs, double* c0, uint32_t nslots = 1, uint32_t c0_stride = 0);
// fp64: residuals of the logged states in half precision, the screening pass's table, the certificate of state 0
hipError_t launch_res_residuals64(ss_hip_ctx* ctx, const double* y, const ResLog<double>& log, double tol, const float* meta, void* r16, float* rn2p,
                                  float* tab, uint32_t* headroom, DevState* st, bool first16, bool omp, uint32_t nslots = 1, const float* slotmeta = nullptr,
                                  uint32_t* fl = nullptr, bool scan = false);

}  // namespace sship
