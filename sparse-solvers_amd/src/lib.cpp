/*
 * lib.cpp — C++14 host library behind include/ss/*.h.  The counterpart of the reference's
 * src/lib.cpp:28-46 (policy -> kernel glue): homotopy_policy::run goes through the compute-mode
 * dispatcher (kernelpp/kernel.h) to solve_homotopy::op<compute_mode::HIP, T> (solvers/homotopy-hip.cpp),
 * which calls the C-ABI of include/ss_hip.h.  HIP is the only mode built: there is no CPU path here — a
 * request for one yields error_code::COMPUTE_MODE_DISABLED, a missing MI355X the error alternative,
 * never a silent fallback.
 */
#include <ss/ss.h>
#include <kernelpp/kernel.h>

#include "solvers/homotopy.h"
#include "ss_hip.h"

#include <cctype>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

namespace kernelpp
{
    /* ---- the compute-mode request (kernelpp/kernel.h) ------------------------------------------------ */
    namespace
    {
        std::mutex& mode_mutex() { static std::mutex mx; return mx; }
        bool g_mode_set = false;
        compute_mode g_mode = compute_mode::AUTO;
    }

    const char* to_string(compute_mode m)
    {
        switch (m) {
            case compute_mode::AUTO: return "AUTO";
            case compute_mode::CPU:  return "CPU";
            case compute_mode::AVX:  return "AVX";
            case compute_mode::HIP:  return "HIP";
        }
        return "?";
    }

    bool parse_mode(const char* text, compute_mode& out)
    {
        if (!text) return false;
        std::string t(text);
        for (char& ch : t) ch = (char)std::toupper((unsigned char)ch);
        if (t == "AUTO" || t.empty()) { out = compute_mode::AUTO; return true; }
        if (t == "HIP" || t == "GPU") { out = compute_mode::HIP; return true; }
        if (t == "CPU") { out = compute_mode::CPU; return true; }
        if (t == "AVX") { out = compute_mode::AVX; return true; }
        return false;
    }

    compute_mode requested_mode()
    {
        std::lock_guard<std::mutex> lock(mode_mutex());
        if (!g_mode_set) {
            compute_mode m = compute_mode::AUTO;
            if (!parse_mode(std::getenv("SS_COMPUTE_MODE"), m)) m = compute_mode::AUTO;
            g_mode = m;
            g_mode_set = true;
        }
        return g_mode;
    }

    void set_requested_mode(compute_mode m)
    {
        std::lock_guard<std::mutex> lock(mode_mutex());
        g_mode = m;
        g_mode_set = true;
    }

    /* HIP: a device is visible to this process.  CPU / AVX: this library has no such implementation. */
    bool mode_available(compute_mode m)
    {
        return m == compute_mode::HIP && ss_hip_device_count() > 0;
    }
}

namespace ss
{
    namespace
    {
        inline ss_hip_ctx* create(const float* A, size_t m, size_t n, ptrdiff_t rs, ptrdiff_t cs,
                                  int dev, char* err, size_t len)
        { return ss_hip_homotopy_create_f32(A, m, n, rs, cs, dev, err, len); }

        inline ss_hip_ctx* create(const double* A, size_t m, size_t n, ptrdiff_t rs, ptrdiff_t cs,
                                  int dev, char* err, size_t len)
        { return ss_hip_homotopy_create_f64(A, m, n, rs, cs, dev, err, len); }

        inline ss_hip_ctx* create_irls(const float* A, size_t m, size_t n, ptrdiff_t rs, ptrdiff_t cs, int dev, char* err, size_t len)
        { return ss_hip_irls_create_f32(A, m, n, rs, cs, dev, err, len); }
        inline ss_hip_ctx* create_irls(const double* A, size_t m, size_t n, ptrdiff_t rs, ptrdiff_t cs, int dev, char* err, size_t len)
        { return ss_hip_irls_create_f64(A, m, n, rs, cs, dev, err, len); }

        /* the mode a state built with `mode` will run in: its own, else the process-wide request */
        inline kernelpp::compute_mode effective(kernelpp::compute_mode mode)
        {
            return mode != kernelpp::compute_mode::AUTO ? mode : kernelpp::requested_mode();
        }
        inline bool wants_device(kernelpp::compute_mode mode)
        {
            const kernelpp::compute_mode e = effective(mode);
            return e == kernelpp::compute_mode::AUTO || e == kernelpp::compute_mode::HIP;
        }
    }

    /* Homotopy solver ----------------------------------------------------- */

    template <typename T>
    void homotopy_state<T>::init(const ndspan<T, 2> A, int device)
    {
        /* a state built without a mode of its own is BOUND to the process-wide request in force now: what is uploaded (or not) here
           and what solve() later asks for cannot drift apart when set_requested_mode() is called in between */
        if (_mode == kernelpp::compute_mode::AUTO) _mode = effective(_mode);
        if (!wants_device(_mode)) {
            /* a mode that is not built: nothing is uploaded; solve() reports COMPUTE_MODE_DISABLED */
            _error = std::string("compute mode ") + kernelpp::to_string(effective(_mode)) + " is not built into this library";
            return;
        }
        char msg[512] = { 0 };
        _ctx = create(A.data(), _m, _n, (ptrdiff_t)A.strides()[0], (ptrdiff_t)A.strides()[1],
                      device, msg, sizeof(msg));
        if (!_ctx) _error = msg;
    }

    template <typename T>
    homotopy_state<T>::homotopy_state(const ndspan<T, 2> A, int device)
        : _ctx(nullptr), _m(A.shape()[0]), _n(A.shape()[1]), _mode(kernelpp::compute_mode::AUTO)
    {
        init(A, device);
    }

    template <typename T>
    homotopy_state<T>::homotopy_state(const ndspan<T, 2> A, kernelpp::compute_mode mode, int device)
        : _ctx(nullptr), _m(A.shape()[0]), _n(A.shape()[1]), _mode(mode)
    {
        init(A, device);
    }

    template <typename T>
    homotopy_state<T>::~homotopy_state()
    {
        if (_ctx) ss_hip_homotopy_destroy(_ctx);
    }

    template class homotopy_state<float>;
    template class homotopy_state<double>;

    /* policy -> kernel glue (reference: src/lib.cpp:30-46, kernelpp::run<solve_homotopy>(...)): the mode is the
       solver's own if it was built with one, else the process-wide request */
    kernelpp::maybe<homotopy_report> homotopy_policy::run(
        homotopy_state<float>& st, const ndspan<float> y, float tol, uint32_t maxiter, ndspan<float> x)
    {
        return kernelpp::run_with<solve_homotopy, kernelpp::maybe<homotopy_report>>(effective(st.mode()), st, y, tol, maxiter, x);
    }

    kernelpp::maybe<homotopy_report> homotopy_policy::run(
        homotopy_state<double>& st, const ndspan<double> y, double tol, uint32_t maxiter, ndspan<double> x)
    {
        return kernelpp::run_with<solve_homotopy, kernelpp::maybe<homotopy_report>>(effective(st.mode()), st, y, tol, maxiter, x);
    }

    kernelpp::maybe<omp_report> omp_policy::run(
        homotopy_state<float>& st, const ndspan<float> y, float tol, uint32_t maxiter, ndspan<float> x)
    {
        return kernelpp::run_with<solve_omp, kernelpp::maybe<omp_report>>(effective(st.mode()), st, y, tol, maxiter, x);
    }

    kernelpp::maybe<omp_report> omp_policy::run(
        homotopy_state<double>& st, const ndspan<double> y, double tol, uint32_t maxiter, ndspan<double> x)
    {
        return kernelpp::run_with<solve_omp, kernelpp::maybe<omp_report>>(effective(st.mode()), st, y, tol, maxiter, x);
    }

    /* IRLS ---------------------------------------------------------------- */

    template <typename T>
    void irls_device_state<T>::init(const ndspan<T, 2> A, int device)
    {
        if (_mode == kernelpp::compute_mode::AUTO) _mode = effective(_mode);      /* (bound to the request in force now: homotopy_state::init) */
        if (!wants_device(_mode)) {
            _error = std::string("compute mode ") + kernelpp::to_string(effective(_mode)) + " is not built into this library";
            return;
        }
        char msg[512] = { 0 };
        _ctx = create_irls(A.data(), _m, _n, (ptrdiff_t)A.strides()[0], (ptrdiff_t)A.strides()[1],
                           device, msg, sizeof(msg));
        if (!_ctx) _error = msg;
    }

    template <typename T>
    irls_device_state<T>::irls_device_state(const ndspan<T, 2> A, int device)
        : _ctx(nullptr), _m(A.shape()[0]), _n(A.shape()[1]), _mode(kernelpp::compute_mode::AUTO)
    {
        init(A, device);
    }

    template <typename T>
    irls_device_state<T>::irls_device_state(const ndspan<T, 2> A, kernelpp::compute_mode mode, int device)
        : _ctx(nullptr), _m(A.shape()[0]), _n(A.shape()[1]), _mode(mode)
    {
        init(A, device);
    }

    template <typename T>
    irls_device_state<T>::~irls_device_state()
    {
        if (_ctx) ss_hip_irls_destroy(_ctx);
    }

    template class irls_device_state<float>;
    template class irls_device_state<double>;

    kernelpp::maybe<irls_report> irls_policy::run(
        irls_device_state<float>& st, const ndspan<float> y, float tol, uint32_t maxiter, ndspan<float> x)
    {
        return kernelpp::run_with<solve_irls, kernelpp::maybe<irls_report>>(effective(st.mode()), st, y, tol, maxiter, x);
    }

    kernelpp::maybe<irls_report> irls_policy::run(
        irls_device_state<double>& st, const ndspan<double> y, double tol, uint32_t maxiter, ndspan<double> x)
    {
        return kernelpp::run_with<solve_irls, kernelpp::maybe<irls_report>>(effective(st.mode()), st, y, tol, maxiter, x);
    }

    /* Utils --------------------------------------------------------------- */

    namespace detail
    {
        inline int hip_reconstruct(ss_hip_ctx* c, const float* x, float* y, char* e, size_t l) { return ss_hip_reconstruct_f32(c, x, y, e, l); }
        inline int hip_reconstruct(ss_hip_ctx* c, const double* x, double* y, char* e, size_t l) { return ss_hip_reconstruct_f64(c, x, y, e, l); }
        inline int hip_norm_l1(float* A, size_t m, size_t n, ptrdiff_t rs, ptrdiff_t cs, char* e, size_t l) { return ss_hip_norm_l1_f32(A, m, n, rs, cs, 0, e, l); }
        inline int hip_norm_l1(double* A, size_t m, size_t n, ptrdiff_t rs, ptrdiff_t cs, char* e, size_t l) { return ss_hip_norm_l1_f64(A, m, n, rs, cs, 0, e, l); }

        /* y = A x on the device (reference: one row-major xgemv, src/lib.cpp:78-92).  Stateless like the
           reference's: the CURRENT contents of the host view are used (no look-up of a solver's device copy —
           a caller may have changed A in place since, e.g. norm_l1(A)).  Only the columns x actually uses
           travel: x is sparse in every use of the reference (test_util.h:167,187), so those columns are
           gathered into a compact matrix, uploaded and multiplied there (ss_hip_reconstruct_*).  No arithmetic
           on the host; the reference's signature cannot report an error, so a failure (no usable GPU) throws. */
        template <typename T>
        void reconstruct_signal(const ndspan<T, 2> A, const ndspan<T> x, ndspan<T> y)
        {
            const size_t m = A.shape()[0], n = A.shape()[1];
            if (x.size() != n || y.size() != m)
                throw std::invalid_argument("reconstruct_signal: vector lengths do not match the shape of A");
            char msg[512] = { 0 };
            std::vector<size_t> nz;
            for (size_t j = 0; j < n; j++) if (x[j] != T(0)) nz.push_back(j);
            if (nz.empty()) {
                for (size_t i = 0; i < m; i++) y[i] = T(0);
                return;
            }
            const size_t k = nz.size();
            std::vector<T> cols(m * k), xk(k), ys(m);
            for (size_t c = 0; c < k; c++) {
                xk[c] = x[nz[c]];
                for (size_t i = 0; i < m; i++) cols[c * m + i] = A(i, nz[c]);       // column-major, compact
            }
            ss_hip_ctx* tmp = create(cols.data(), m, k, (ptrdiff_t)1, (ptrdiff_t)m, 0, msg, sizeof(msg));
            if (!tmp) throw std::runtime_error(std::string("reconstruct_signal: ") + msg);
            const int rc = hip_reconstruct(tmp, xk.data(), ys.data(), msg, sizeof(msg));
            ss_hip_homotopy_destroy(tmp);
            if (rc != SS_HIP_OK) throw std::runtime_error(std::string("reconstruct_signal: ") + msg);
            for (size_t i = 0; i < m; i++) y[i] = ys[i];
        }

        /* every column of A divided by its l1 norm, in place, on the device (ss_hip_norm_l1_*:
           reference src/linalg/norms.h:22-27); a failure (no usable GPU) throws */
        template <typename T>
        void norm_l1(ndspan<T, 2> A)
        {
            char msg[512] = { 0 };
            if (A.shape()[0] == 0 || A.shape()[1] == 0) return;
            const int rc = hip_norm_l1(A.data(), A.shape()[0], A.shape()[1], (ptrdiff_t)A.strides()[0],
                                       (ptrdiff_t)A.strides()[1], msg, sizeof(msg));
            if (rc != SS_HIP_OK) throw std::runtime_error(std::string("norm_l1: ") + msg);
        }
    }

    void reconstruct_signal(const ndspan<float, 2> A, const ndspan<float> x, ndspan<float> y) {
        detail::reconstruct_signal(A, x, y);
    }

    void reconstruct_signal(const ndspan<double, 2> A, const ndspan<double> x, ndspan<double> y) {
        detail::reconstruct_signal(A, x, y);
    }

    void norm_l1(ndspan<float, 2> A)  { detail::norm_l1(A); }
    void norm_l1(ndspan<double, 2> A) { detail::norm_l1(A); }
}
