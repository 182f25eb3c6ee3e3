/*
 * ss/ndspan.h — non-owning strided n-dimensional view, the argument type at the drop-in
 * boundary.  Mirrors the names of the reference's include/ss/ndspan.h:28-165 (there an
 * xtensor adaptor; xtensor is not a dependency here): ndspan<T, N>, the as_span(...)
 * overload set and view().
 */
#pragma once

#include <array>
#include <cstddef>
#include <type_traits>

namespace ss
{
    /*
     *  A non-owning n-dimensional view over a (ptr, shape, strides) representation;
     *  strides are in ELEMENTS.  Row-major by default.  (reference: ndspan.h:28-31)
     */
    template <typename T, size_t NDim = 1>
    class ndspan
    {
      public:
        using value_type   = T;
        using shape_type   = std::array<size_t, NDim>;
        using strides_type = std::array<size_t, NDim>;

        ndspan() : _data(nullptr), _shape{}, _strides{} {}

        ndspan(T* data, const shape_type& shape) : _data(data), _shape(shape) {
            size_t s = 1;
            for (size_t d = NDim; d-- > 0;) { _strides[d] = s; s *= _shape[d]; }
        }

        ndspan(T* data, const shape_type& shape, const strides_type& strides)
            : _data(data), _shape(shape), _strides(strides) {}

        /* views of const data convert from views of mutable data */
        template <typename U, typename = typename std::enable_if<
            std::is_same<const U, T>::value && !std::is_same<U, T>::value>::type>
        ndspan(const ndspan<U, NDim>& o) : _data(o.data()), _shape(o.shape()), _strides(o.strides()) {}

        const shape_type&   shape()   const { return _shape; }
        const strides_type& strides() const { return _strides; }
        size_t dimension() const { return NDim; }

        size_t size() const {
            size_t s = 1;
            for (size_t d = 0; d < NDim; d++) s *= _shape[d];
            return s;
        }

        T* data() const { return _data; }
        /* xtensor-style accessors used by reference call sites (blas_wrapper.h:63-70) */
        T* raw_data() const { return _data; }
        size_t raw_data_offset() const { return 0; }

        template <typename... I>
        T& operator()(I... idx) const {
            static_assert(sizeof...(I) == NDim, "wrong number of indices");
            const size_t ii[] = { static_cast<size_t>(idx)... };
            size_t off = 0;
            for (size_t d = 0; d < NDim; d++) off += ii[d] * _strides[d];
            return _data[off];
        }

        template <size_t M = NDim>
        typename std::enable_if<M == 1, T&>::type operator[](size_t i) const {
            return _data[i * _strides[0]];
        }

      private:
        T*           _data;
        shape_type   _shape;
        strides_type _strides;
    };


    /* as_span ------------------------------------------------------------- */

    /* n-d view of the given shape over an stl-like container (ndspan.h:56-64) */
    template <size_t N, typename C>
    ndspan<typename C::value_type, N> as_span(C& container, std::array<size_t, N> shape) {
        return ndspan<typename C::value_type, N>(container.data(), shape);
    }

    /* 1-d view of a (ptr, len) representation (ndspan.h:69-79) */
    template <typename T>
    ndspan<T, 1> as_span(T* buf, size_t len) {
        return ndspan<T, 1>(buf, std::array<size_t, 1>{ { len } });
    }

    template <typename T>
    const ndspan<T, 1> as_span(const T* buf, size_t len) {
        return as_span<T>(const_cast<T*>(buf), len);
    }

    /* n-d view of the given shape over a pointer (ndspan.h:85-98) */
    template <size_t N, typename T>
    ndspan<T, N> as_span(T* buf, std::array<size_t, N> shape) {
        return ndspan<T, N>(buf, shape);
    }

    template <size_t N, typename T>
    const ndspan<T, N> as_span(const T* buf, std::array<size_t, N> shape) {
        return as_span<N, T>(const_cast<T*>(buf), shape);
    }

    /* n-d view with per-dimension strides in elements (ndspan.h:105-118) */
    template <size_t N, typename T>
    ndspan<T, N> as_span(T* buf, std::array<size_t, N> shape, std::array<size_t, N> strides) {
        return ndspan<T, N>(buf, shape, strides);
    }

    template <size_t N, typename T>
    const ndspan<T, N> as_span(const T* buf, std::array<size_t, N> shape, std::array<size_t, N> strides) {
        return as_span<N, T>(const_cast<T*>(buf), shape, strides);
    }

    /* 1-d view of an stl-like container (ndspan.h:146-156) */
    template <typename C>
    inline auto as_span(C& container)
        -> ndspan<typename std::remove_pointer<decltype(container.data())>::type, 1>
    {
        using V = typename std::remove_pointer<decltype(container.data())>::type;
        return ndspan<V, 1>(container.data(), std::array<size_t, 1>{ { container.size() } });
    }


    /* view ---------------------------------------------------------------- */

    /*
     *  Assignable whole-span proxy: `ss::view(x) = T(0)`, `ss::view(x) = other`
     *  (the subset of xt::view(e, xt::all()) the reference's call sites use,
     *  ndspan.h:161-165, test_util.h:42-45).
     */
    template <typename T>
    struct span_view
    {
        ndspan<T, 1> s;

        span_view& operator=(T v) {
            for (size_t i = 0; i < s.size(); i++) s[i] = v;
            return *this;
        }
        template <typename U>
        span_view& operator=(const ndspan<U, 1>& o) {
            for (size_t i = 0; i < s.size(); i++) s[i] = o[i];
            return *this;
        }
        span_view& operator/=(T v) {
            for (size_t i = 0; i < s.size(); i++) s[i] /= v;
            return *this;
        }
    };

    template <typename T>
    inline span_view<T> view(ndspan<T, 1> s) { return span_view<T>{ s }; }
}
