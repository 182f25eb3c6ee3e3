/*
 * ss/ss.h — the C++14 entry point of the MI355X-native sparse solvers.
 *
 * Drop-in for the solver front-end of the reference library (include/ss/ss.h:25-115): the
 * same class template ss::solver<T, Policy>, the same aliases and free functions, so code
 * written against the reference recompiles against this header and links
 * libsparsesolvers.so instead.  What differs is underneath: constructing a solver uploads the
 * sensing matrix to the GPU once, and solve() runs the device-resident iteration through the
 * C-ABI of include/ss_hip.h.
 *
 *     std::vector<float> A(m * n), y(m), x(n);
 *     ss::homotopy<float> solver(ss::as_span<2>(A.data(), { m, n }));
 *     auto result = solver.solve(ss::as_span(y), 1e-3f, 256, ss::as_span(x));
 *     if (result.is<ss::homotopy_report>()) { ... result.get<ss::homotopy_report>().iter ... }
 */
#pragma once

#include "ss/fwd.h"
#include "ss/ndspan.h"
#include "ss/policies.h"

#include <kernelpp/types.h>

#include <cstdint>
#include <memory>

namespace ss
{
    /*
     * solver<T, Policy>: T is float or double; Policy supplies
     *     report_type                       what a successful solve returns
     *     state_type<T>                     what is built once from the matrix
     *     run(state, y, tol, max_it, x)     one solve
     * The object is movable, not copyable (it owns device memory through its state).
     */
    template <typename T, typename Policy>
    struct solver
    {
        using report_type  = typename Policy::report_type;
        using state_type   = typename Policy::template state_type<T>;
        using solve_result = kernelpp::maybe<report_type>;

        static_assert(detail::is_solver<Policy, T>::value,
                      "Policy::run(state_type<T>&, ndspan<T>, T, integer, ndspan<T>) is required");

        /* Captures the m x n sensing matrix viewed by A (any element strides).  The data is
           copied to the device here; the caller's buffer is not referenced afterwards. */
        solver(const ndspan<T, 2> A) : m(new state_type(A)) {}

        /* The same with an explicit compute mode for this solver (the reference picks the mode inside
           kernelpp::run, src/lib.cpp:36; here it can also be pinned per solver).  AUTO = the process-wide
           request (environment variable SS_COMPUTE_MODE, else the best mode available); a mode this library
           is not built with — it has a HIP back-end only — makes every solve return
           kernelpp::error_code::COMPUTE_MODE_DISABLED.  For policies whose state takes a mode. */
        solver(const ndspan<T, 2> A, kernelpp::compute_mode mode) : m(new state_type(A, mode)) {}

        solver(solver&& other) : m(std::move(other.m)) {}

        ~solver() = default;

        /*
         * min ||x||_1  subject to  A x = y      (policy-specific relaxation / stopping rule)
         *
         *   y               the signal, length m
         *   tol             stop once the residual correlation ||A^T (y - A x)||_inf <= tol
         *   max_iterations  iteration budget, > 0
         *   x               receives the sparse representation, length n (fully overwritten)
         *
         * Returns the policy's report, or a kernelpp::error describing what went wrong
         * (bad arguments, no usable GPU, ...) — errors are values, nothing is thrown.
         */
        solve_result solve(const ndspan<T> y, T tol, std::uint32_t max_iterations, ndspan<T> x)
        {
            return Policy::run(*m, y, tol, max_iterations, x);
        }

      private:
        std::unique_ptr<state_type> m;
    };

    /* l1 homotopy (the reference's ss::homotopy) */
    template <typename T> using homotopy = solver<T, homotopy_policy>;

    /* orthogonal matching pursuit — an addition, the reference has no OMP */
    template <typename T> using omp = solver<T, omp_policy>;

    /* iteratively reweighted least squares (reference: ss.h:63-64); rows(A) >= columns(A) */
    template <typename T> using irls = solver<T, irls_policy>;


    /* y = A x: rebuilds a signal from its sparse representation (reference: ss.h:67-83).
       A is m x n, x has n entries, y receives m entries. */
    void reconstruct_signal(const ndspan<float, 2> A, const ndspan<float> x, ndspan<float> y);
    void reconstruct_signal(const ndspan<double, 2> A, const ndspan<double> x, ndspan<double> y);

    /* Scales every column of A, in place, to unit l1 norm (reference: ss.h:86-93). */
    void norm_l1(ndspan<float, 2> A);
    void norm_l1(ndspan<double, 2> A);
}
