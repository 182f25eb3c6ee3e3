/*
 * kernelpp/types.h — the result / error vocabulary of the drop-in boundary.
 *
 * The reference returns `kernelpp::maybe<report>` from solver<T,P>::solve
 * (include/ss/ss.h:32) and its call sites use exactly
 *     r.is<R>()  r.get<R>()  r.get_unchecked<R>()  r.is<kernelpp::error>()
 *     r.get<kernelpp::error>().data()                       (homotopy_test.cpp:12-13,
 *     homotopy_bench.cpp:45, binding.cpp:48-55)
 * and `kernelpp::status` (true on error, .get().data(): binding.cpp:57-62).  kernelpp is
 * an un-vendored submodule of the reference; this header provides those names with
 * those members so that code written against the reference compiles unchanged.
 */
#pragma once

#include <stdexcept>
#include <string>
#include <utility>

namespace kernelpp
{
    enum class compute_mode { AUTO = 0, CPU, AVX, HIP };

    enum class error_code { NONE = 0, KERNEL_FAILED, COMPUTE_MODE_DISABLED, INVALID_ARGUMENT };

    /* an error message; `.data()` yields a NUL-terminated string */
    class error
    {
      public:
        error() : _code(error_code::KERNEL_FAILED) {}
        error(std::string msg, error_code c = error_code::KERNEL_FAILED)
            : _msg(std::move(msg)), _code(c) {}
        error(error_code c) : _msg(c == error_code::NONE ? "" : "kernel failed"), _code(c) {}

        const char* data() const { return _msg.c_str(); }
        const std::string& str() const { return _msg; }
        error_code code() const { return _code; }

      private:
        std::string _msg;
        error_code  _code;
    };

    inline bool operator==(const error& a, const error& b) { return a.str() == b.str(); }

    /* report-or-error */
    template <typename R>
    class maybe
    {
      public:
        maybe(const R& r) : _ok(true), _value(r) {}
        maybe(R&& r) : _ok(true), _value(std::move(r)) {}
        maybe(const error& e) : _ok(false), _value(), _error(e) {}
        maybe(error&& e) : _ok(false), _value(), _error(std::move(e)) {}
        maybe(error_code c) : _ok(c == error_code::NONE), _value(), _error(c) {}

        template <typename U> bool is() const { return is_impl(static_cast<U*>(nullptr)); }

        template <typename U> const U& get() const {
            if (!is<U>()) throw std::runtime_error("kernelpp::maybe: bad access");
            return get_impl(static_cast<U*>(nullptr));
        }
        template <typename U> U& get() {
            if (!is<U>()) throw std::runtime_error("kernelpp::maybe: bad access");
            return const_cast<U&>(get_impl(static_cast<U*>(nullptr)));
        }
        template <typename U> const U& get_unchecked() const { return get_impl(static_cast<U*>(nullptr)); }

        explicit operator bool() const { return _ok; }

      private:
        bool is_impl(R*) const { return _ok; }
        bool is_impl(error*) const { return !_ok; }
        const R& get_impl(R*) const { return _value; }
        const error& get_impl(error*) const { return _error; }

        bool  _ok;
        R     _value;
        error _error;
    };

    /* success-or-error; converts to TRUE when it holds an error */
    class status
    {
      public:
        status() : _failed(false) {}
        status(const error& e) : _failed(true), _error(e) {}
        status(error_code c) : _failed(c != error_code::NONE), _error(c) {}

        explicit operator bool() const { return _failed; }
        const error& get() const { return _error; }

      private:
        bool  _failed;
        error _error;
    };
}
