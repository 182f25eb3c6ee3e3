// subcheck.h — the reference's predicates for ONE (column outside the subset, logged breakpoint) of a subset solve: shared by the
// subset form's check over all columns (subbatch.hip: k_sub_verify) and the screened form's exact re-check of the columns its
// half-precision certificate could not vouch for (screen.hip: k_scr_recheck).
#pragma once

#include "ss_hip_internal.h"
#include "ss_hip_device.h"

namespace sship {

// the reference's predicates for one (column, breakpoint): cv, qv by the chain; h = the breakpoint's header
__device__ __forceinline__ void sub_check(float cv, float qv, uint32_t j, uint32_t k, uint32_t nlog, const uint32_t* sH, float tol,
                                          int tie_guard, DevState* st, bool& fail, bool& tie)
{
    const uint32_t* h = sH + k * 8;
    const float lam = __uint_as_float(h[4]);
    const float ac = fabsf(cv);
    if (!(h[1] & 1u)) {
        // the round the path ended in: ||c||_inf is REPORTED (homotopy_report::solution_error) and, unless the budget ran
        // out, was compared with the tolerance — a larger |c| out here must leave that comparison as it was, and counts
        if (!(ac <= lam)) {
            if (!(lam > tol) && !(ac <= tol)) fail = true;
            else if (ac == ac) atomicMax(reinterpret_cast<unsigned long long*>(&st->c_inf), (unsigned long long)__double_as_longlong((double)ac));
            else fail = true;
        }
        return;
    }
    if (!(ac <= lam)) fail = true;                             // the true max |c| is larger (or NaN): every candidate changes
    const float gam = __uint_as_float(h[5]);
    const float dl = 1.f - qv, dr = 1.f + qv;
    const float nl = lam - cv, nr = lam + cv;
    // safe: the candidate is certainly larger than the step taken (1e-4 covers every rounding in between)
    const float bound = gam * 1.0001f;
    const bool safe_l = dl > 0.f && nl > dl * bound;
    const bool safe_r = dr > 0.f && nr > dr * bound;
    if (safe_l && safe_r) return;
    const uint32_t pick = h[2];
    const bool in_band = h[7] != 0u;
    const uint32_t jr = h[6];
    float m = Lim<float>::max();
    if (dl != 0.f) {
        float t = nl / dl;
        if (tie_guard && t == 0.f && dl > 0.f) t = Lim<float>::tiny();
        if (t == 0.f && j != jr && in_band) tie = true;
        if (t > 0.f && t < m) m = t;
    }
    if (dr != 0.f) {
        float t = nr / dr;
        if (tie_guard && t == 0.f && dr > 0.f) t = Lim<float>::tiny();
        if (t == 0.f && j != jr && in_band) tie = true;
        if (t > 0.f && t < m) m = t;
    }
    if (better_min(m, j, gam, pick)) {
        // it would have been picked instead.  One exception: the LAST step of a path that ends by tolerance — there
        // lambda - gamma ~ 0 and every column's candidate ties with the step within rounding; whichever is
        // inserted enters with x = 0 and the next round ends the path with the same coefficients.  A candidate
        // within 1e-5 of the step taken is such a tie, a smaller one a real entrant.
        // (and taking the shorter step must still end the path: lambda after it is the logged one plus the difference)
        const float lam_end = __uint_as_float(sH[(k + 1u) * 8 + 4]);
        const bool last_step = k + 2u == nlog && !(sH[(k + 1u) * 8 + 1] & 1u) && !(lam_end > tol);
        if (!(last_step && m >= gam * 0.99999f && lam_end + (gam - m) <= tol)) fail = true;
    }
}

// The same predicates with the state's figures handed over as values (fp64 logs keep lambda and the step in doubles beside the
// header): lam / gam = lambda and the step of state k, lam_end / end_is_final = lambda of state k + 1 and whether the path ends there.
template <typename T>
__device__ __forceinline__ void sub_check_v(T cv, T qv, uint32_t j, bool has_scan, T lam, T gam, uint32_t pick, bool in_band, uint32_t jr,
                                            bool is_last_scan, T lam_end, bool end_is_final, T tol, int tie_guard, DevState* st, bool& fail, bool& tie)
{
    const T ac = cv < T(0) ? -cv : cv;
    if (!has_scan) {
        if (!(ac <= lam)) {
            if (!(lam > tol) && !(ac <= tol)) fail = true;
            else if (ac == ac) atomicMax(reinterpret_cast<unsigned long long*>(&st->c_inf), (unsigned long long)__double_as_longlong((double)ac));
            else fail = true;
        }
        return;
    }
    if (!(ac <= lam)) fail = true;
    const T dl = T(1) - qv, dr = T(1) + qv;
    const T nl = lam - cv, nr = lam + cv;
    const T bound = gam * T(1.0001);                              // (only a shortcut: what it cannot rule out is evaluated exactly below)
    const bool safe_l = dl > T(0) && nl > dl * bound;
    const bool safe_r = dr > T(0) && nr > dr * bound;
    if (safe_l && safe_r) return;
    T m = Lim<T>::max();
    if (dl != T(0)) {
        T t = nl / dl;
        if (tie_guard && t == T(0) && dl > T(0)) t = Lim<T>::tiny();
        if (t == T(0) && j != jr && in_band) tie = true;
        if (t > T(0) && t < m) m = t;
    }
    if (dr != T(0)) {
        T t = nr / dr;
        if (tie_guard && t == T(0) && dr > T(0)) t = Lim<T>::tiny();
        if (t == T(0) && j != jr && in_band) tie = true;
        if (t > T(0) && t < m) m = t;
    }
    if (better_min(m, j, gam, pick)) {
        const bool last_step = is_last_scan && end_is_final && !(lam_end > tol);
        // (the window of "a tie of all columns within rounding" scales with the precision: 1e-5 in fp32, 1e-12 in fp64)
        const T window = sizeof(T) == 8 ? T(1) - T(1e-12) : T(0.99999);
        if (!(last_step && m >= gam * window && lam_end + (gam - m) <= tol)) fail = true;
    }
}

}  // namespace sship
