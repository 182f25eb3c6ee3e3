// C++14 user of the drop-in API (include/ss/ss.h), restating the reference's
// src/solvers/homotopy_test.cpp:8-40 + src/solvers/test_util.h:27-92 against ss::homotopy<T>.
//   test_ss_api                 -> needs an MI355X; exit code 0 when every check passes
//   test_ss_api --no-device     -> checks the error convention when no GPU is usable
#include <ss/ss.h>
#include <kernelpp/kernel.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <cmath>
#include <vector>

static int failures = 0;
#define CHECK(cond)                                                                    \
    do {                                                                               \
        if (!(cond)) { std::printf("CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); ++failures; } \
    } while (0)

template <typename Report>
void check_report(kernelpp::maybe<Report>& result, float tolerance, uint32_t max_iterations)
{
    CHECK(result.template is<Report>());
    if (!result.template is<Report>()) {
        std::printf("  error: %s\n", result.template get<kernelpp::error>().data());
        return;
    }
    auto r = result.template get<Report>();
    CHECK(r.iter >= 1);
    CHECK(r.iter <= max_iterations);
    if (r.iter < max_iterations) CHECK(r.solution_error <= tolerance);
}

template <typename T>
void smoke_test()
{
    const uint32_t N = 5;
    std::vector<T> identity(N * N, T(0)), signal(N), x(N);
    for (uint32_t i = 0; i < N; i++) identity[i * N + i] = T(1);

    ss::homotopy<T> solver(ss::as_span<2>(identity.data(), { N, N }));
    for (uint32_t n = 0; n < N; n++) {
        ss::view(ss::as_span(signal)) = T(0);
        signal[n] = T(1);
        ss::view(ss::as_span(x)) = T(0);
        auto result = solver.solve(ss::as_span(signal), T(.001), N, ss::as_span(x));
        check_report(result, .001f, N);
        CHECK(x == signal);
    }
}

template <typename T>
void smoke_test_column_subset()
{
    const size_t N = 10, M = 5;
    std::vector<T> data(M * N, T(0));
    for (size_t i = 0; i < M; i++) {
        for (size_t j = 0; j + 1 < M; j++) data[i * N + j] = T(0.01) * T((i * 7 + j * 3) % 10);
        data[i * N + M + i] = T(1);
    }
    // columns 5..9 of the 5 x 10 buffer: row stride 10
    auto identity = ss::as_span<2>(data.data() + M, { M, M }, { N, size_t(1) });
    ss::homotopy<T> solver(identity);
    std::vector<T> signal(M), x(M);
    for (size_t n = 0; n < M; n++) {
        for (size_t i = 0; i < M; i++) signal[i] = identity(i, n);
        ss::view(ss::as_span(x)) = T(0);
        auto result = solver.solve(ss::as_span(signal), T(.001), uint32_t(N), ss::as_span(x));
        CHECK(result.template is<ss::homotopy_report>());
        CHECK(x == signal);
    }
}

void error_convention()
{
    std::vector<float> A(16, 0.f), y(4, 1.f), x(4, 0.f);
    for (int i = 0; i < 4; i++) A[i * 4 + i] = 1.f;
    ss::homotopy<float> s(ss::as_span<2>(A.data(), { size_t(4), size_t(4) }));
    ss::homotopy<float> moved(std::move(s));                      // move-only, like the reference
    auto bad = moved.solve(ss::as_span(y), 0.5f, 0, ss::as_span(x));   // max_iterations == 0
    CHECK(bad.is<kernelpp::error>());
    CHECK(!bad.is<ss::homotopy_report>());
    CHECK(std::strlen(bad.get<kernelpp::error>().data()) > 0);
    auto bad2 = moved.solve(ss::as_span(y), 1.5f, 4, ss::as_span(x));  // tol >= 1
    CHECK(bad2.is<kernelpp::error>());
    std::vector<float> yshort(3, 1.f);
    auto bad3 = moved.solve(ss::as_span(yshort), 0.5f, 4, ss::as_span(x));
    CHECK(bad3.is<kernelpp::error>());
    auto ok = moved.solve(ss::as_span(y), 0.001f, 8, ss::as_span(x));
    CHECK(ok.is<ss::homotopy_report>());
    CHECK(ok.get_unchecked<ss::homotopy_report>().iter >= 1);
}

void omp_api()
{
    // ss::omp<T>: greedy pursuit on the same solver<T, Policy> template
    const size_t M = 6, N = 6;
    std::vector<double> A(M * N, 0.0), y(M, 0.0), x(N, 0.0);
    for (size_t i = 0; i < M; i++) A[i * N + i] = 1.0;
    y[1] = 2.0; y[4] = -3.0;
    ss::omp<double> solver(ss::as_span<2>(A.data(), { M, N }));
    auto r = solver.solve(ss::as_span(y), 1e-9, 6, ss::as_span(x));
    CHECK(r.is<ss::omp_report>());
    if (r.is<ss::omp_report>()) {
        CHECK(r.get<ss::omp_report>().iter == 2);
        CHECK(r.get<ss::omp_report>().solution_error <= 1e-9);
    }
    CHECK(x == y);
}

void irls_api()
{
    // ss::irls<T> (reference: ss.h:63-64, irls_test.cpp smoke_test): identity => x == y, one iteration
    const size_t N = 5;
    std::vector<float> A(N * N, 0.f);
    for (size_t i = 0; i < N; i++) A[i * N + i] = 1.f;
    ss::irls<float> solver(ss::as_span<2>(A.data(), { N, N }));
    for (size_t n = 0; n < N; n++) {
        std::vector<float> y(N, 0.f), x(N, -1.f);
        y[n] = 1.f;
        auto r = solver.solve(ss::as_span(y), 0.001f, 5, ss::as_span(x));
        CHECK(r.is<ss::irls_report>());
        if (r.is<ss::irls_report>()) {
            const auto rep = r.get<ss::irls_report>();
            CHECK(rep.iter == 1 && !rep.spd_failure && rep.solution_error == 0.0);
        }
        CHECK(x == y);
    }
    // rows < columns is a construction error, reported by solve() (no exceptions)
    std::vector<double> W(2 * 3, 1.0), yw(2, 1.0), xw(3, 0.0);
    ss::irls<double> wide(ss::as_span<2>(W.data(), { size_t(2), size_t(3) }));
    CHECK(wide.solve(ss::as_span(yw), 0.1, 3, ss::as_span(xw)).is<kernelpp::error>());
}

void utilities()
{
    // norm_l1 literal (reference: src/linalg/norms_test.cpp:10-26): columns divided by their l1 norms
    {
        std::vector<float> A = { 1, 2, 0,
                                 3, 4, 1 };   // 2 x 3
        ss::norm_l1(ss::as_span<2>(A.data(), { size_t(2), size_t(3) }));
        const float expect[6] = { 0.25f, 0.3333f, 0.f, 0.75f, 0.6667f, 1.f };
        for (int e = 0; e < 6; e++) CHECK(std::fabs(A[e] - expect[e]) <= 1e-4f);
    }
    // exact values, a strided (column-subset) view and a column-major view
    std::vector<double> A = { 1, 2, 3, 6 };   // 2 x 2
    ss::norm_l1(ss::as_span<2>(A.data(), { size_t(2), size_t(2) }));
    CHECK(A[0] == 0.25 && A[1] == 0.25 && A[2] == 0.75 && A[3] == 0.75);
    {
        std::vector<double> W = { 9, 1, 2, 9,
                                  9, 3, 6, 9 };   // columns 1..2 of a 2 x 4 buffer
        ss::norm_l1(ss::as_span<2>(W.data() + 1, { size_t(2), size_t(2) }, { size_t(4), size_t(1) }));
        CHECK(W[1] == 0.25 && W[2] == 0.25 && W[5] == 0.75 && W[6] == 0.75 && W[0] == 9 && W[3] == 9 && W[7] == 9);
        std::vector<double> C = { 1, 3, 2, 6 };   // the same matrix, column-major
        ss::norm_l1(ss::as_span<2>(C.data(), { size_t(2), size_t(2) }, { size_t(1), size_t(2) }));
        CHECK(C[0] == 0.25 && C[1] == 0.75 && C[2] == 0.25 && C[3] == 0.75);
    }
    // reconstruct_signal: no solver holds A -> the used columns travel to the device
    std::vector<double> x = { 2, 0 }, y(2, -1);
    ss::reconstruct_signal(ss::as_span<2>(A.data(), { size_t(2), size_t(2) }), ss::as_span(x), ss::as_span(y));
    CHECK(y[0] == 0.5 && y[1] == 1.5);
    // ... stateless like the reference's: with a solver alive on the same view, a later in-place change of the host
    // matrix IS seen (the solver keeps its own device copy; reconstruct_signal reads the caller's current A)
    {
        std::vector<double> B = A;
        ss::homotopy<double> solver(ss::as_span<2>(B.data(), { size_t(2), size_t(2) }));
        std::vector<double> x2 = { 0, 4 }, y2(2, -1);
        ss::reconstruct_signal(ss::as_span<2>(B.data(), { size_t(2), size_t(2) }), ss::as_span(x2), ss::as_span(y2));
        CHECK(y2[0] == 1.0 && y2[1] == 3.0);
        B[1] = 0.5;                                                   // A(0, 1): 0.25 -> 0.5
        ss::reconstruct_signal(ss::as_span<2>(B.data(), { size_t(2), size_t(2) }), ss::as_span(x2), ss::as_span(y2));
        CHECK(y2[0] == 2.0 && y2[1] == 3.0);
    }
    std::vector<double> xz = { 0, 0 }, yz(2, -1);
    ss::reconstruct_signal(ss::as_span<2>(A.data(), { size_t(2), size_t(2) }), ss::as_span(xz), ss::as_span(yz));
    CHECK(yz[0] == 0.0 && yz[1] == 0.0);
}

// The compute-mode seam (kernelpp/kernel.h; reference: src/solvers/homotopy.h:27-38 KERNEL_DECL + op<mode, T>,
// src/lib.cpp:36 kernelpp::run): this library is built with the HIP mode only.  A solver pinned to a mode that
// is not built, or a process-wide request for one, reports error_code::COMPUTE_MODE_DISABLED — with or without a GPU.
void compute_mode_seam(bool have_device)
{
    using kernelpp::compute_mode;
    using kernelpp::error_code;
    std::vector<float> A = { 1, 0, 0, 1 }, y = { 1, 0 }, x(2);
    const auto Av = ss::as_span<2>(A.data(), { size_t(2), size_t(2) });
    compute_mode parsed = compute_mode::AUTO;
    CHECK(kernelpp::parse_mode("hip", parsed) && parsed == compute_mode::HIP);
    CHECK(kernelpp::parse_mode("CPU", parsed) && parsed == compute_mode::CPU);
    CHECK(!kernelpp::parse_mode("tpu", parsed));
    CHECK(kernelpp::mode_available(compute_mode::HIP) == have_device);
    CHECK(!kernelpp::mode_available(compute_mode::CPU) && !kernelpp::mode_available(compute_mode::AVX));
    for (compute_mode m : { compute_mode::CPU, compute_mode::AVX }) {
        ss::homotopy<float> pinned(Av, m);
        auto r = pinned.solve(ss::as_span(y), 0.001f, 2, ss::as_span(x));
        CHECK(r.is<kernelpp::error>());
        if (r.is<kernelpp::error>()) CHECK(r.get<kernelpp::error>().code() == error_code::COMPUTE_MODE_DISABLED);
        ss::omp<float> pinned_omp(Av, m);
        auto ro = pinned_omp.solve(ss::as_span(y), 0.001f, 2, ss::as_span(x));
        CHECK(ro.is<kernelpp::error>() && ro.get<kernelpp::error>().code() == error_code::COMPUTE_MODE_DISABLED);
        ss::irls<float> pinned_irls(Av, m);
        auto ri = pinned_irls.solve(ss::as_span(y), 0.001f, 2, ss::as_span(x));
        CHECK(ri.is<kernelpp::error>() && ri.get<kernelpp::error>().code() == error_code::COMPUTE_MODE_DISABLED);
    }
    // the process-wide request (what SS_COMPUTE_MODE sets at first use)
    const compute_mode before = kernelpp::requested_mode();
    kernelpp::set_requested_mode(compute_mode::CPU);
    {
        ss::homotopy<float> s(Av);
        auto r = s.solve(ss::as_span(y), 0.001f, 2, ss::as_span(x));
        CHECK(r.is<kernelpp::error>() && r.get<kernelpp::error>().code() == error_code::COMPUTE_MODE_DISABLED);
        // a solver pinned to HIP is not affected by the process-wide request
        ss::homotopy<float> hip(Av, compute_mode::HIP);
        auto rh = hip.solve(ss::as_span(y), 0.001f, 2, ss::as_span(x));
        if (have_device) { CHECK(rh.is<ss::homotopy_report>()); CHECK(x == y); }
        else { CHECK(rh.is<kernelpp::error>() && rh.get<kernelpp::error>().code() != error_code::COMPUTE_MODE_DISABLED); }
    }
    // a solver built without a mode of its own is BOUND to the request in force at its construction: one built while the request
    // was CPU (nothing uploaded) still answers COMPUTE_MODE_DISABLED after the request has moved on to HIP — not a half-made state
    {
        ss::homotopy<float> made_under_cpu(Av);
        kernelpp::set_requested_mode(compute_mode::HIP);
        auto r = made_under_cpu.solve(ss::as_span(y), 0.001f, 2, ss::as_span(x));
        CHECK(r.is<kernelpp::error>() && r.get<kernelpp::error>().code() == error_code::COMPUTE_MODE_DISABLED);
        kernelpp::set_requested_mode(compute_mode::CPU);
    }
    kernelpp::set_requested_mode(compute_mode::HIP);
    {
        ss::homotopy<float> s(Av);
        auto r = s.solve(ss::as_span(y), 0.001f, 2, ss::as_span(x));
        if (have_device) CHECK(r.is<ss::homotopy_report>());
        else CHECK(r.is<kernelpp::error>() && r.get<kernelpp::error>().code() == error_code::KERNEL_FAILED);
    }
    kernelpp::set_requested_mode(before);
}

int no_device()
{
    std::vector<float> A = { 1, 0, 0, 1 }, y = { 1, 0 }, x(2);
    ss::homotopy<float> s(ss::as_span<2>(A.data(), { size_t(2), size_t(2) }));
    auto r = s.solve(ss::as_span(y), 0.001f, 2, ss::as_span(x));
    CHECK(r.is<kernelpp::error>());
    if (r.is<kernelpp::error>()) std::printf("error (expected): %s\n", r.get<kernelpp::error>().data());
    // the utilities run on the device as well: without one they fail loudly (no host fallback)
    bool threw = false;
    try { ss::norm_l1(ss::as_span<2>(A.data(), { size_t(2), size_t(2) })); } catch (const std::runtime_error& e) { threw = true; std::printf("error (expected): %s\n", e.what()); }
    CHECK(threw && A[0] == 1.f);
    threw = false;
    try { ss::reconstruct_signal(ss::as_span<2>(A.data(), { size_t(2), size_t(2) }), ss::as_span(y), ss::as_span(x)); } catch (const std::runtime_error& e) { threw = true; }
    CHECK(threw);
    compute_mode_seam(false);
    return failures;
}

int main(int argc, char** argv)
{
    static_assert(ss::detail::is_solver<ss::homotopy_policy, float>::value, "f32");
    static_assert(ss::detail::is_solver<ss::homotopy_policy, double>::value, "f64");
    static_assert(ss::detail::is_solver<ss::omp_policy, float>::value, "omp f32");
    static_assert(ss::detail::is_solver<ss::irls_policy, double>::value, "irls f64");
    if (argc > 1 && !std::strcmp(argv[1], "--no-device")) return no_device() ? 1 : 0;
    smoke_test<float>();
    smoke_test<double>();
    smoke_test_column_subset<float>();
    smoke_test_column_subset<double>();
    error_convention();
    omp_api();
    irls_api();
    utilities();
    compute_mode_seam(true);
    std::printf("%s (%d failures)\n", failures ? "FAILED" : "ok", failures);
    return failures ? 1 : 0;
}
