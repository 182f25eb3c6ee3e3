/*
 * ss/policies.h — solver policies (reference: include/ss/policies.h:21-54).
 *
 * homotopy_policy keeps the reference's interface (report_type, state_type<T>, run x
 * {float,double}); its state is no longer the bare view of A but an owning object that
 * holds the MI355X-resident copy of the sensing matrix (the policy design allows a
 * non-trivial state: the reference's own irls_state, policies.h:77-85).  The copy is made
 * when the solver is constructed; later mutation of the caller's A is not observed.
 */
#pragma once

#include "ss/ndspan.h"
#include <kernelpp/types.h>

#include <cstdint>
#include <string>

struct ss_hip_ctx;   /* include/ss_hip.h */

namespace ss
{
    /* Homotopy ------------------------------------------------------------ */

    struct homotopy_report
    {
        /* The number of iterations performed. */
        uint32_t iter;

        /* The solution error */
        double solution_error;
    };

    inline bool operator== (const homotopy_report&, const homotopy_report&) { return false; }

    /* Device-side state of one homotopy solver: the context of include/ss_hip.h */
    template <typename T>
    class homotopy_state
    {
      public:
        /* uploads (and re-lays-out) the m x n view A to HIP device `device` */
        explicit homotopy_state(const ndspan<T, 2> A, int device = 0);
        /* the same with an explicit compute mode for this solver (kernelpp/kernel.h): AUTO and HIP upload;
           a mode this library is not built with (CPU, AVX) uploads nothing and every solve returns
           error_code::COMPUTE_MODE_DISABLED */
        homotopy_state(const ndspan<T, 2> A, kernelpp::compute_mode mode, int device = 0);
        ~homotopy_state();

        homotopy_state(const homotopy_state&) = delete;
        homotopy_state& operator=(const homotopy_state&) = delete;

        ss_hip_ctx* ctx() const { return _ctx; }
        size_t rows() const { return _m; }
        size_t cols() const { return _n; }
        /* non-empty when construction failed (no device, out of memory, ...) */
        const std::string& error() const { return _error; }
        /* the compute mode asked for at construction (AUTO: the process-wide request decides at solve time) */
        kernelpp::compute_mode mode() const { return _mode; }

      private:
        void init(const ndspan<T, 2> A, int device);
        ss_hip_ctx* _ctx;
        size_t      _m, _n;
        std::string _error;
        kernelpp::compute_mode _mode;
    };

    /* Orthogonal matching pursuit ------------------------------------------ */

    /* Not part of the reference (which ships homotopy and irls only, ss.h:60-64); same
       report shape as homotopy_report: iterations and ||A^T r||_inf at exit. */
    struct omp_report
    {
        uint32_t iter;
        double solution_error;
    };

    inline bool operator== (const omp_report&, const omp_report&) { return false; }

    /* greedy l0 pursuit on the same device-resident matrix copy as homotopy_policy */
    struct omp_policy
    {
        using report_type = omp_report;

        template <typename T> using state_type = homotopy_state<T>;

        static kernelpp::maybe<omp_report> run(
            state_type<float>&, const ndspan<float>, float, uint32_t, ndspan<float>);

        static kernelpp::maybe<omp_report> run(
            state_type<double>&, const ndspan<double>, double, uint32_t, ndspan<double>);
    };

    /* IRLS ---------------------------------------------------------------- */

    /* reference: include/ss/policies.h:58-72 */
    struct irls_report
    {
        /* The number of iterations performed. */
        uint32_t iter;

        /* The solution error */
        double solution_error;

        /* Whether IRLS stopped because an iteration met a matrix that is not symmetric
           positive definite (no full Cholesky decomposition). */
        bool spd_failure;
    };

    inline bool operator== (const irls_report&, const irls_report&) { return false; }

    /* Device-side state of one IRLS solver (reference: irls_state holds the QR decomposition of
       A, policies.h:77-85): the factorisation lives on the MI355X.  Needs rows >= columns. */
    template <typename T>
    class irls_device_state
    {
      public:
        explicit irls_device_state(const ndspan<T, 2> A, int device = 0);
        irls_device_state(const ndspan<T, 2> A, kernelpp::compute_mode mode, int device = 0);
        ~irls_device_state();

        irls_device_state(const irls_device_state&) = delete;
        irls_device_state& operator=(const irls_device_state&) = delete;

        ss_hip_ctx* ctx() const { return _ctx; }
        size_t rows() const { return _m; }
        size_t cols() const { return _n; }
        const std::string& error() const { return _error; }
        kernelpp::compute_mode mode() const { return _mode; }

      private:
        void init(const ndspan<T, 2> A, int device);
        ss_hip_ctx* _ctx;
        size_t      _m, _n;
        std::string _error;
        kernelpp::compute_mode _mode;
    };

    /* A solver policy which implements the Iteratively Reweighted Least Squares method
       (reference: policies.h:88-100) */
    struct irls_policy
    {
        using report_type = irls_report;

        template <typename T> using state_type = irls_device_state<T>;

        static kernelpp::maybe<irls_report> run(
            state_type<float>&, const ndspan<float>, float, uint32_t, ndspan<float>);

        static kernelpp::maybe<irls_report> run(
            state_type<double>&, const ndspan<double>, double, uint32_t, ndspan<double>);
    };

    /* A solver policy which implements the homotopy method on an MI355X */
    struct homotopy_policy
    {
        using report_type = homotopy_report;

        template <typename T> using state_type = homotopy_state<T>;

        static kernelpp::maybe<homotopy_report> run(
            state_type<float>&, const ndspan<float>, float, uint32_t, ndspan<float>);

        static kernelpp::maybe<homotopy_report> run(
            state_type<double>&, const ndspan<double>, double, uint32_t, ndspan<double>);
    };
}
