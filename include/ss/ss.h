/*
 * ss/ss.h — public C++14 API, drop-in for the Homotopy path of the reference
 * (include/ss/ss.h:25-115): ss::solver<T, Policy>, ss::homotopy<T>, reconstruct_signal,
 * norm_l1.  Same names, signatures and error convention; link libsparsesolvers.so.
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
    /* Solver base --------------------------------------------------------- */

    template <typename T, typename SolverPolicy>
    struct solver
    {
        using report_type  = typename SolverPolicy::report_type;
        using state_type   = typename SolverPolicy::template state_type<T>;
        using solve_result = kernelpp::maybe<report_type>;

        /* A : view of a sensing matrix (copied to the device here) */
        solver(const ndspan<T, 2> A);

        ~solver() = default;

        /*  Uses the SolverPolicy to solve the equation
         *    min || x || _1  subject to A x = y
         *
         *                 y : signal vector of length m
         *    max_iterations : maximum number of iterations
         *               tol : sparsity budget
         *                 x : the output sparse representation vector
         *                     of length n
         *
         *    returns : an instance of report_type, or an error
         */
        solve_result solve(const ndspan<T> y, T tol, std::uint32_t max_iterations, ndspan<T> x);

        solver(solver<T, SolverPolicy>&& other) : m{ std::move(other.m) } {}

      private:
        std::unique_ptr<state_type> m;
    };

    /* Solver types  ------------------------------------------------------- */

    template <typename T>
    using homotopy = solver<T, homotopy_policy>;

    /* orthogonal matching pursuit (an addition: the reference has no OMP) */
    template <typename T>
    using omp = solver<T, omp_policy>;


    /* Utilities ----------------------------------------------------------- */

    /*  computes A x : reconstructs a signal from its sparse representation
     *  (reference: ss.h:67-83, lib.cpp:78-104)
     */
    void reconstruct_signal(
        const ndspan<float, 2> A, const ndspan<float> x, ndspan<float> y);

    void reconstruct_signal(
        const ndspan<double, 2> A, const ndspan<double> x, ndspan<double> y);

    /*  Normalizes the columns of A in place by their L1 norm
     *  (reference: ss.h:86-93, norms.h:22-27)
     */
    void norm_l1(ndspan<float, 2> A);

    void norm_l1(ndspan<double, 2> A);


    /* Definitions --------------------------------------------------------- */

    template <typename T, typename S>
    solver<T, S>::solver(const ndspan<T, 2> A)
        : m(new state_type(A))
    {
        static_assert(
            detail::is_solver<S, T>::value,
            "The specified solver policy does not implment the required interface");
    }

    template <typename T, typename S>
    typename solver<T, S>::solve_result solver<T, S>::solve(
        const ndspan<T>     y,
              T             tolerance,
              std::uint32_t max_iterations,
              ndspan<T>     x)
    {
        return S::run(*m, y, tolerance, max_iterations, x);
    }
}
