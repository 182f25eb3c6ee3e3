/*
 * ss/fwd.h — is_solver trait (reference: include/ss/fwd.h:19-38): well formed when the
 * policy P can solve problems parameterised by T, i.e. provides
 *     P::run(P::state_type<T>&, ndspan<T>, T, size_t, ndspan<T>).
 */
#pragma once

#include "ss/ndspan.h"

#include <type_traits>
#include <utility>

namespace ss {
    namespace detail
    {
        using std::declval;

        template <typename...> struct make_void { using type = void; };
        template <typename... Ts> using void_t = typename make_void<Ts...>::type;

        template <typename P, typename T>
        using solvable = decltype(
            P::run(declval<typename P::template state_type<T>&>(),
                   declval<ndspan<T>>(), T{0}, std::size_t{0},
                   declval<ndspan<T>>()));

        template <typename P, typename T, typename = void>
        struct is_solver : std::false_type {};

        template <typename P, typename T>
        struct is_solver <P, T, void_t<solvable<P, T>>> : std::true_type {};
    }
}
