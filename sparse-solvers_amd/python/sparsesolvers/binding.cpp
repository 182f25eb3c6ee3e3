/*
 * binding.cpp — pybind11 module `sparsesolvers.binding`, the Python surface of the
 * reference (bindings/python/sparsesolvers/binding.cpp:114-148) for the Homotopy path:
 *   version() -> [major, minor, patch]
 *   HomotopyReport{iter, solution_error}
 *   Homotopy(A: ndarray[f32|f64, 2-D, any strides])
 *   Homotopy.solve(b, tolerance=eps(T)*10, max_iterations=100) -> (x, HomotopyReport)
 * Errors surface as RuntimeError.  The sensing matrix is copied to the MI355X when the
 * solver is constructed (so, unlike the reference, no dangling view of A is kept); the GIL
 * is released while the device solves.
 */
#include <pybind11/pybind11.h>
#include <pybind11/numpy.h>
#include <pybind11/stl.h>

#include <ss/ss.h>

#include <array>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

namespace py = pybind11;

namespace
{
    /* API level of the reference this module mirrors (sparse-solvers v0.8.8) */
    const int VERSION[3] = { 0, 8, 8 };

    /* numpy array -> strided view (element strides); the dimension-count message is the
       reference's (binding.cpp:24-25) because user code may match on it */
    template <size_t Rank, typename T>
    ss::ndspan<T, Rank> view_of(py::array_t<T>& array)
    {
        const py::ssize_t have = array.ndim();
        if (have != (py::ssize_t)Rank) {
            throw std::runtime_error("Unexpected number of dimensions. Expected "
                                     + std::to_string(Rank) + " but got " + std::to_string(have));
        }
        std::array<size_t, Rank> extent, step;
        for (size_t axis = 0; axis < Rank; ++axis) {
            const py::ssize_t bytes = array.strides((py::ssize_t)axis);
            if (bytes < 0) {
                throw std::runtime_error("Negative strides are not supported; pass numpy.ascontiguousarray(a)");
            }
            extent[axis] = (size_t)array.shape((py::ssize_t)axis);
            step[axis] = (size_t)bytes / sizeof(T);
        }
        return ss::ndspan<T, Rank>(array.mutable_data(), extent, step);
    }

    /* library errors are values; Python gets them as RuntimeError */
    template <typename Report>
    void raise_if_error(const kernelpp::maybe<Report>& outcome)
    {
        if (!outcome.template is<Report>())
            throw std::runtime_error(outcome.template get<kernelpp::error>().data());
    }

    /* one Python class per policy, holding a float OR a double solver (binding.cpp:64-72) */
    template <typename Policy>
    struct py_solver
    {
        std::array<size_t, 2> shape;
        std::unique_ptr<ss::solver<float, Policy>>  f32;
        std::unique_ptr<ss::solver<double, Policy>> f64;
    };
    using py_homotopy = py_solver<ss::homotopy_policy>;
    using py_omp = py_solver<ss::omp_policy>;
    using py_irls = py_solver<ss::irls_policy>;

    template <typename T, typename P> struct slot_of;
    template <typename P> struct slot_of<float, P>  { static std::unique_ptr<ss::solver<float, P>>&  get(py_solver<P>& s) { return s.f32; } };
    template <typename P> struct slot_of<double, P> { static std::unique_ptr<ss::solver<double, P>>& get(py_solver<P>& s) { return s.f64; } };

    template <typename T, typename P>
    void def_init(py::class_<py_solver<P>>& cls)
    {
        cls.def(py::init([](py::array_t<T> A_) {
            auto A = view_of<2>(A_);
            auto* self = new py_solver<P>{ A.shape(), nullptr, nullptr };
            slot_of<T, P>::get(*self).reset(new ss::solver<T, P>(A));
            return self;
        }), py::arg("A"));
    }

    template <typename T, typename P>
    void def_solve(py::class_<py_solver<P>>& cls)
    {
        using report_type = typename P::report_type;
        cls.def("solve",
            [](py_solver<P>& self, py::array_t<T> b, T tol, uint32_t maxiter)
            {
                auto& s = slot_of<T, P>::get(self);
                if (!s) throw std::runtime_error(
                    "dtype of b does not match the dtype of the sensing matrix");
                py::array_t<T> x((py::ssize_t)self.shape[1]);
                auto bs = view_of<1>(b);
                auto xs = view_of<1>(x);
                kernelpp::maybe<report_type> result = report_type{};
                {
                    py::gil_scoped_release release;
                    result = s->solve(bs, tol, maxiter, xs);
                }
                raise_if_error(result);
                return std::make_tuple(x, result.template get<report_type>());
            },
            "Execute the solver on the given inputs.",
            py::arg("b").noconvert(),
            py::arg("tolerance") = std::numeric_limits<T>::epsilon() * 10,
            py::arg("max_iterations") = 100);
    }
}

PYBIND11_MODULE(binding, m)
{
    m.doc() = "MI355X-native sparse-solvers (Homotopy l1) binding";
    m.def("version", []() { return std::vector<int>(VERSION, VERSION + 3); },
          "API version of sparsesolvers this module mirrors");

    /* homotopy report */
    py::class_<ss::homotopy_report>(m, "HomotopyReport")
        .def(py::init([]() { return ss::homotopy_report{ 0u, 0.0 }; }))
        .def_readwrite("iter", &ss::homotopy_report::iter)
        .def_readwrite("solution_error", &ss::homotopy_report::solution_error);

    /* homotopy solver */
    auto homotopy = py::class_<py_homotopy>(m, "Homotopy");
    def_init<float, ss::homotopy_policy>(homotopy);
    def_init<double, ss::homotopy_policy>(homotopy);
    def_solve<float, ss::homotopy_policy>(homotopy);
    def_solve<double, ss::homotopy_policy>(homotopy);

    /* irls report and solver (reference: binding.cpp:133-146) */
    py::class_<ss::irls_report>(m, "IrlsReport")
        .def(py::init([]() { return ss::irls_report{ 0u, 0.0, false }; }))
        .def_readwrite("iter", &ss::irls_report::iter)
        .def_readwrite("spd_failure", &ss::irls_report::spd_failure)
        .def_readwrite("solution_error", &ss::irls_report::solution_error);
    auto irls = py::class_<py_irls>(m, "Irls");
    def_init<float, ss::irls_policy>(irls);
    def_init<double, ss::irls_policy>(irls);
    def_solve<float, ss::irls_policy>(irls);
    def_solve<double, ss::irls_policy>(irls);

    /* orthogonal matching pursuit (an addition; the reference exposes Homotopy and Irls) */
    py::class_<ss::omp_report>(m, "OmpReport")
        .def(py::init([]() { return ss::omp_report{ 0u, 0.0 }; }))
        .def_readwrite("iter", &ss::omp_report::iter)
        .def_readwrite("solution_error", &ss::omp_report::solution_error);
    auto omp = py::class_<py_omp>(m, "Omp");
    def_init<float, ss::omp_policy>(omp);
    def_init<double, ss::omp_policy>(omp);
    def_solve<float, ss::omp_policy>(omp);
    def_solve<double, ss::omp_policy>(omp);
}
