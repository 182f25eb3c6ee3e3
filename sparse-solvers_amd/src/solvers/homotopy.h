/*
 * solvers/homotopy.h — kernel declarations of the MI355X back-end.
 *
 * Counterpart of the reference's src/solvers/homotopy.h:27-38 and src/solvers/irls.h:27-38: the same
 * KERNEL_DECL shape, declared for ONE compute mode, HIP.  The specialisations op<compute_mode::HIP, T>
 * (homotopy-hip.cpp) forward to the C-ABI of include/ss_hip.h; a maintainer of the reference would add
 * exactly these to its own declaration list.  Differences from the reference's
 * signature: the first argument is the policy's state (the device-resident copy of A made at solver
 * construction) instead of the bare view of A, and the result carries a message next to the error code.
 */
#pragma once

#include <ss/ss.h>
#include <kernelpp/kernel.h>

namespace ss
{
    using kernelpp::compute_mode;
    using kernelpp::error_code;

    KERNEL_DECL(solve_homotopy, compute_mode::HIP)
    {
        template <compute_mode, typename T>
        static kernelpp::maybe<homotopy_report> op(
            homotopy_state<T>& state, const ndspan<T> y, T tolerance, std::uint32_t max_iterations, ndspan<T> x);
    };

    /* the specialisations homotopy-hip.cpp defines, DECLARED before any use that would instantiate the primary template
       ([temp.expl.spec]/6: without these the program is ill-formed, no diagnostic required — it only linked because the
       primary has no definition) */
    template <> kernelpp::maybe<homotopy_report> solve_homotopy::op<compute_mode::HIP, float>(
        homotopy_state<float>&, const ndspan<float>, float, std::uint32_t, ndspan<float>);
    template <> kernelpp::maybe<homotopy_report> solve_homotopy::op<compute_mode::HIP, double>(
        homotopy_state<double>&, const ndspan<double>, double, std::uint32_t, ndspan<double>);

    /* orthogonal matching pursuit: not in the reference (include/ss/ss.h:60-64) */
    KERNEL_DECL(solve_omp, compute_mode::HIP)
    {
        template <compute_mode, typename T>
        static kernelpp::maybe<omp_report> op(
            homotopy_state<T>& state, const ndspan<T> y, T tolerance, std::uint32_t max_iterations, ndspan<T> x);
    };

    template <> kernelpp::maybe<omp_report> solve_omp::op<compute_mode::HIP, float>(
        homotopy_state<float>&, const ndspan<float>, float, std::uint32_t, ndspan<float>);
    template <> kernelpp::maybe<omp_report> solve_omp::op<compute_mode::HIP, double>(
        homotopy_state<double>&, const ndspan<double>, double, std::uint32_t, ndspan<double>);

    /* reference: src/solvers/irls.h:27-38 */
    KERNEL_DECL(solve_irls, compute_mode::HIP)
    {
        template <compute_mode, typename T>
        static kernelpp::maybe<irls_report> op(
            irls_device_state<T>& state, const ndspan<T> y, T tolerance, std::uint32_t max_iterations, ndspan<T> x);
    };

    template <> kernelpp::maybe<irls_report> solve_irls::op<compute_mode::HIP, float>(
        irls_device_state<float>&, const ndspan<float>, float, std::uint32_t, ndspan<float>);
    template <> kernelpp::maybe<irls_report> solve_irls::op<compute_mode::HIP, double>(
        irls_device_state<double>&, const ndspan<double>, double, std::uint32_t, ndspan<double>);
}
