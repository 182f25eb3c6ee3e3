/*
 * solvers/homotopy-hip.cpp — op<compute_mode::HIP, T> of the kernels declared in solvers/homotopy.h.
 *
 * Counterpart of the reference's src/solvers/homotopy-cpu.cpp:277-297 (op<compute_mode::CPU, float/double>
 * -> run_solver<T>) and src/solvers/irls-cpu.cpp:127-146: each specialisation checks the argument shapes the
 * reference asserts (homotopy-cpu.cpp:193-199) and forwards to the C-ABI of include/ss_hip.h, where the
 * device-resident iteration runs.  Nothing is computed on the host.
 */
#include "solvers/homotopy.h"

#include "ss_hip.h"

#include <cstddef>
#include <string>

namespace ss
{
    namespace
    {
        inline int c_solve(ss_hip_ctx* c, const float* y, ptrdiff_t incy, float tol, uint32_t it,
                           float* x, ptrdiff_t incx, uint32_t* io, double* eo, char* err, size_t len)
        { return ss_hip_homotopy_solve_f32(c, y, incy, tol, it, x, incx, io, eo, err, len); }
        inline int c_solve(ss_hip_ctx* c, const double* y, ptrdiff_t incy, double tol, uint32_t it,
                           double* x, ptrdiff_t incx, uint32_t* io, double* eo, char* err, size_t len)
        { return ss_hip_homotopy_solve_f64(c, y, incy, tol, it, x, incx, io, eo, err, len); }

        inline int c_omp(ss_hip_ctx* c, const float* y, ptrdiff_t incy, float tol, uint32_t it,
                         float* x, ptrdiff_t incx, uint32_t* io, double* eo, char* err, size_t len)
        { return ss_hip_omp_solve_f32(c, y, incy, tol, it, x, incx, io, eo, err, len); }
        inline int c_omp(ss_hip_ctx* c, const double* y, ptrdiff_t incy, double tol, uint32_t it,
                         double* x, ptrdiff_t incx, uint32_t* io, double* eo, char* err, size_t len)
        { return ss_hip_omp_solve_f64(c, y, incy, tol, it, x, incx, io, eo, err, len); }

        inline int c_irls(ss_hip_ctx* c, const float* y, ptrdiff_t incy, float tol, uint32_t it, float* x, ptrdiff_t incx,
                          uint32_t* io, double* eo, int* spd, char* err, size_t len)
        { return ss_hip_irls_solve_f32(c, y, incy, tol, it, x, incx, io, eo, spd, err, len); }
        inline int c_irls(ss_hip_ctx* c, const double* y, ptrdiff_t incy, double tol, uint32_t it, double* x, ptrdiff_t incx,
                          uint32_t* io, double* eo, int* spd, char* err, size_t len)
        { return ss_hip_irls_solve_f64(c, y, incy, tol, it, x, incx, io, eo, spd, err, len); }

        inline kernelpp::error failure(int rc, const char* msg)
        {
            return kernelpp::error(msg, rc == SS_HIP_EINVAL ? error_code::INVALID_ARGUMENT : error_code::KERNEL_FAILED);
        }

        template <typename State, typename T>
        inline bool shapes_ok(const State& st, const ndspan<T>& y, const ndspan<T>& x)
        {
            return y.size() == st.rows() && x.size() == st.cols();
        }
    }

    /* ---- Homotopy ------------------------------------------------------------------------------------- */
    template <typename T>
    static kernelpp::maybe<homotopy_report> homotopy_hip(
        homotopy_state<T>& st, const ndspan<T> y, T tol, std::uint32_t maxiter, ndspan<T> x)
    {
        if (!st.ctx())
            return kernelpp::error(st.error().empty() ? "homotopy: no device context" : st.error());
        if (y.size() != st.rows())
            return kernelpp::error("homotopy: length of y does not match the rows of A", error_code::INVALID_ARGUMENT);
        if (x.size() != st.cols())
            return kernelpp::error("homotopy: length of x does not match the columns of A", error_code::INVALID_ARGUMENT);
        char msg[512] = { 0 };
        homotopy_report rep{ 0u, 0.0 };
        const int rc = c_solve(st.ctx(), y.data(), (ptrdiff_t)y.strides()[0], tol, maxiter,
                               x.data(), (ptrdiff_t)x.strides()[0], &rep.iter, &rep.solution_error, msg, sizeof(msg));
        if (rc != SS_HIP_OK) return failure(rc, msg);
        return rep;
    }

    template <> kernelpp::maybe<homotopy_report> solve_homotopy::op<compute_mode::HIP, float>(
        homotopy_state<float>& st, const ndspan<float> y, float tol, std::uint32_t maxiter, ndspan<float> x)
    {
        return homotopy_hip<float>(st, y, tol, maxiter, x);
    }

    template <> kernelpp::maybe<homotopy_report> solve_homotopy::op<compute_mode::HIP, double>(
        homotopy_state<double>& st, const ndspan<double> y, double tol, std::uint32_t maxiter, ndspan<double> x)
    {
        return homotopy_hip<double>(st, y, tol, maxiter, x);
    }

    /* ---- OMP ------------------------------------------------------------------------------------------ */
    template <typename T>
    static kernelpp::maybe<omp_report> omp_hip(
        homotopy_state<T>& st, const ndspan<T> y, T tol, std::uint32_t maxiter, ndspan<T> x)
    {
        if (!st.ctx())
            return kernelpp::error(st.error().empty() ? "omp: no device context" : st.error());
        if (!shapes_ok(st, y, x))
            return kernelpp::error("omp: vector lengths do not match the shape of A", error_code::INVALID_ARGUMENT);
        char msg[512] = { 0 };
        omp_report rep{ 0u, 0.0 };
        const int rc = c_omp(st.ctx(), y.data(), (ptrdiff_t)y.strides()[0], tol, maxiter,
                             x.data(), (ptrdiff_t)x.strides()[0], &rep.iter, &rep.solution_error, msg, sizeof(msg));
        if (rc != SS_HIP_OK) return failure(rc, msg);
        return rep;
    }

    template <> kernelpp::maybe<omp_report> solve_omp::op<compute_mode::HIP, float>(
        homotopy_state<float>& st, const ndspan<float> y, float tol, std::uint32_t maxiter, ndspan<float> x)
    {
        return omp_hip<float>(st, y, tol, maxiter, x);
    }

    template <> kernelpp::maybe<omp_report> solve_omp::op<compute_mode::HIP, double>(
        homotopy_state<double>& st, const ndspan<double> y, double tol, std::uint32_t maxiter, ndspan<double> x)
    {
        return omp_hip<double>(st, y, tol, maxiter, x);
    }

    /* ---- IRLS ----------------------------------------------------------------------------------------- */
    template <typename T>
    static kernelpp::maybe<irls_report> irls_hip(
        irls_device_state<T>& st, const ndspan<T> y, T tol, std::uint32_t maxiter, ndspan<T> x)
    {
        if (!st.ctx())
            return kernelpp::error(st.error().empty() ? "irls: no device context" : st.error());
        if (!shapes_ok(st, y, x))
            return kernelpp::error("irls: vector lengths do not match the shape of A", error_code::INVALID_ARGUMENT);
        char msg[512] = { 0 };
        irls_report rep{ 0u, 0.0, false };
        int spd = 0;
        const int rc = c_irls(st.ctx(), y.data(), (ptrdiff_t)y.strides()[0], tol, maxiter,
                              x.data(), (ptrdiff_t)x.strides()[0], &rep.iter, &rep.solution_error, &spd, msg, sizeof(msg));
        if (rc != SS_HIP_OK) return failure(rc, msg);
        rep.spd_failure = spd != 0;
        return rep;
    }

    template <> kernelpp::maybe<irls_report> solve_irls::op<compute_mode::HIP, float>(
        irls_device_state<float>& st, const ndspan<float> y, float tol, std::uint32_t maxiter, ndspan<float> x)
    {
        return irls_hip<float>(st, y, tol, maxiter, x);
    }

    template <> kernelpp::maybe<irls_report> solve_irls::op<compute_mode::HIP, double>(
        irls_device_state<double>& st, const ndspan<double> y, double tol, std::uint32_t maxiter, ndspan<double> x)
    {
        return irls_hip<double>(st, y, tol, maxiter, x);
    }
}
