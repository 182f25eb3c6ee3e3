/*
 * kernelpp/kernel.h — the compute-mode seam of the drop-in boundary.
 *
 * The reference declares a kernel with the compute modes it is built for and specialises
 * `op<compute_mode, T>` per mode (src/solvers/homotopy.h:27-38: KERNEL_DECL(solve_homotopy,
 * compute_mode::CPU); src/solvers/homotopy-cpu.cpp:277-297: op<compute_mode::CPU, float/double>);
 * `kernelpp::run<K>(args...)` (src/lib.cpp:36,45) then picks, at run time, the best declared mode the
 * machine supports (src/linalg/blas_wrapper.cpp:27-31,55-66 shows the pattern with AVX and CPU).
 * kernelpp is an un-vendored submodule of the reference; this header states that seam with the same
 * names so that the MI355X back-end is what SURVEY §1 calls it: "one more compute_mode + one more op<>
 * specialisation" (sparse-solvers_amd/src/solvers/homotopy.h, homotopy-hip.cpp).
 *
 * Mode selection:
 *   - a kernel lists its modes in order of preference: KERNEL_DECL(solve_homotopy, compute_mode::HIP)
 *   - the request is compute_mode::AUTO unless overridden — process-wide by the environment variable
 *     SS_COMPUTE_MODE (AUTO | HIP | CPU | AVX, read once) or kernelpp::set_requested_mode(), per solver by
 *     the constructor argument of ss::solver<T, P>
 *   - AUTO runs the first declared mode that is available on this machine
 *   - an explicit request for a mode the kernel does not declare — this library is built with HIP only:
 *     there is NO CPU implementation in it — yields error_code::COMPUTE_MODE_DISABLED, never a silent
 *     substitute; a declared mode without a usable device yields KERNEL_FAILED with a message.
 */
#pragma once

#include <kernelpp/types.h>

#include <utility>

namespace kernelpp
{
    template <compute_mode... Ms> struct mode_list {};

    /* base of every kernel declaration: the modes it has op<> specialisations for */
    template <typename K, compute_mode... Ms>
    struct kernel
    {
        using modes = mode_list<Ms...>;
    };

#define KERNEL_DECL(NAME, ...) struct NAME : ::kernelpp::kernel<NAME, __VA_ARGS__>

    /* implemented in libsparsesolvers.so (src/lib.cpp) */
    compute_mode requested_mode();                 /* SS_COMPUTE_MODE at first use, AUTO if unset / unknown */
    void         set_requested_mode(compute_mode);
    bool         mode_available(compute_mode);     /* HIP: a gfx device is visible; CPU / AVX: not built -> false */
    const char*  to_string(compute_mode);
    bool         parse_mode(const char* text, compute_mode& out);

    namespace detail
    {
        template <typename K, typename R>
        struct runner
        {
            template <typename... Args>
            static R go(compute_mode want, bool& declared, mode_list<>, Args&&...)
            {
                if (want != compute_mode::AUTO && !declared)
                    return R(error(std::string("compute mode ") + to_string(want) + " is not built into this library",
                                   error_code::COMPUTE_MODE_DISABLED));
                return R(error(std::string("no HIP device available (compute mode ") +
                               (want == compute_mode::AUTO ? "AUTO" : to_string(want)) + ")", error_code::KERNEL_FAILED));
            }

            template <compute_mode M, compute_mode... Rest, typename... Args>
            static R go(compute_mode want, bool& declared, mode_list<M, Rest...>, Args&&... args)
            {
                if (want == compute_mode::AUTO || want == M) {
                    declared = true;
                    if (mode_available(M)) return K::template op<M>(std::forward<Args>(args)...);
                }
                return go(want, declared, mode_list<Rest...>{}, std::forward<Args>(args)...);
            }
        };
    }

    /* run kernel K in the requested mode (AUTO: the first declared mode this machine supports) */
    template <typename K, typename R, typename... Args>
    R run_with(compute_mode want, Args&&... args)
    {
        bool declared = false;
        return detail::runner<K, R>::go(want, declared, typename K::modes{}, std::forward<Args>(args)...);
    }

    /* the reference's entry: the process-wide request (src/lib.cpp:36,45) */
    template <typename K, typename R, typename... Args>
    R run(Args&&... args)
    {
        return run_with<K, R>(requested_mode(), std::forward<Args>(args)...);
    }
}
