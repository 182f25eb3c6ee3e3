/*
 * ss/fwd.h — compile-time check that a policy type can drive ss::solver<T, Policy>.
 *
 * A policy qualifies for element type T when the expression
 *     Policy::run(state, y, tolerance, max_iterations, x)
 * is valid for a `Policy::state_type<T>& state` and 1-d views y, x of T.  This is the role
 * of the reference's detail::is_solver (include/ss/fwd.h:19-38), written here as an
 * overload-resolution probe instead of a void_t partial specialisation.
 */
#pragma once

#include "ss/ndspan.h"

#include <cstddef>
#include <type_traits>
#include <utility>

namespace ss { namespace detail {

    struct solver_probe
    {
        /* preferred overload: participates only when Policy::run(...) is well formed */
        template <typename Policy, typename Elem,
                  typename State = typename Policy::template state_type<Elem>,
                  typename = decltype(Policy::run(std::declval<State&>(),
                                                  std::declval<ndspan<Elem>>(),
                                                  Elem(),
                                                  std::size_t(),
                                                  std::declval<ndspan<Elem>>()))>
        static std::true_type test(int);

        /* fallback */
        template <typename Policy, typename Elem>
        static std::false_type test(...);
    };

    /* is_solver<Policy, T>::value — true when Policy can solve problems over T */
    template <typename Policy, typename Elem>
    struct is_solver : decltype(solver_probe::test<Policy, Elem>(0)) {};

}}
