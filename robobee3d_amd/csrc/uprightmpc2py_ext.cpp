// Compiled Python module of the drop-in boundary: the counterpart of the reference's pybind11 extension
// template/uprightmpc2/py/uprightmpc2py.cpp:30-80 (classes UprightMPC2C and WLCon over umpcInit / umpcUpdate and
// wlConInit / wlConUpdate), here over libumpc_mi355x.so. The reference converts its arguments through pybind11/eigen.h
// (Eigen is not in this image and is not needed: nothing here computes); this module takes the same numpy arrays through
// pybind11/numpy.h -- a row-major 3 x 3 R0 becomes the column-major matrix umpcUpdate expects (uprightmpc2.c:219), exactly
// what the Eigen caster does -- and returns the same tuples of float32 arrays. Built by robobee3d_amd/_lib.py::build_ext()
// with the host compiler (no device code); robobee3d_amd/uprightmpc2py.py prefers it over its ctypes classes, which
// remain as the portable binding of the same C symbols. Extensions beyond the reference's class: an
// optional actualT0 (the reference's own harness calls update with six arguments, template/uprightmpc2.py:139), status(),
// set_compat().
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <cstring>
#include <stdexcept>
#include <tuple>

#include "umpc_mi355x.h"

namespace py = pybind11;
using farr = py::array_t<float, py::array::c_style | py::array::forcecast>;

namespace {

const float *vec(const farr &a, py::ssize_t n, const char *name) {
  if (a.size() != n) throw std::invalid_argument(std::string(name) + ": wrong number of elements");
  return a.data();
}
farr out(const float *p, py::ssize_t n) {
  farr r(n);
  std::memcpy(r.mutable_data(), p, (size_t)n * sizeof(float));
  return r;
}

class UprightMPC2 {
 public:
  UprightMPC_t umpc;

  UprightMPC2(float dt, float g, float TtoWmax, float ws, float wds, float wpr, float wpf, float wvr, float wvf, float wthrust,
              float wmom, const farr &Ib, int maxIter) {
    umpcInit(&umpc, dt, g, TtoWmax, ws, wds, wpr, wpf, wvr, wvf, wthrust, wmom, vec(Ib, 3, "Ib"), maxIter);
  }
  UprightMPC2(const UprightMPC2 &) = delete;
  UprightMPC2 &operator=(const UprightMPC2 &) = delete;
  ~UprightMPC2() { umpcRelease(&umpc); }

  std::tuple<farr, farr> update(const farr &p0, const farr &R0, const farr &dq0, const farr &pdes, const farr &dpdes, const farr &sdes,
                                float actualT0) {
    const float *R = vec(R0, 9, "R0");
    float Rc[9];
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Rc[3 * c + r] = R[3 * r + c];      // row-major numpy -> column-major (matmult.h:20)
    float uquad[3], accdes[6];
    int rc;
    {
      py::gil_scoped_release nogil;      // (the reference holds the GIL; releasing it costs nothing and lets hosts thread)
      rc = umpcUpdate(&umpc, uquad, accdes, vec(p0, 3, "p0"), Rc, vec(dq0, 6, "dq0"), vec(pdes, 3, "pdes"), vec(dpdes, 3, "dpdes"),
                      vec(sdes, 3, "sdes"), actualT0);
    }
    if (rc) throw std::runtime_error("umpcUpdate failed (no GPU / not initialised)");
    return std::make_tuple(out(uquad, 3), out(accdes, 6));
  }
  std::tuple<farr, farr, farr> vectors() { return std::make_tuple(out(umpc.l, UMPC_NC), out(umpc.u, UMPC_NC), out(umpc.q, UMPC_NX)); }
  std::tuple<farr, farr, py::array_t<int>> matrices() {
    py::array_t<int> idx(UMPC_nAdata);
    std::memcpy(idx.mutable_data(), umpc.Ax_idx, UMPC_nAdata * sizeof(int));
    return std::make_tuple(out(umpc.Px_data, UMPC_NX), out(umpc.Ax_data, UMPC_nAdata), idx);
  }
  int status() { return umpcLastStatus(&umpc); }
  int set_compat(int flags) {
    const int rc = umpcSetCompat(&umpc, flags);
    if (rc < 0) throw std::runtime_error("umpcSetCompat: no live controller");
    return rc;
  }
  float T0() const { return umpc.T0; }
  void set_T0(float v) { umpc.T0 = v; }
};

class WLCon {
 public:
  WLCon_t wl;
  WLCon(const farr &u0, const farr &umin, const farr &umax, const farr &dumax, const farr &Qw, float controlRate, const farr &popts) {
    wlConInit(&wl, vec(u0, 4, "u0"), vec(umin, 4, "umin"), vec(umax, 4, "umax"), vec(dumax, 4, "dumax"), vec(Qw, 6, "Qw"), controlRate,
              vec(popts, 90, "popts"));
  }
  std::tuple<farr, farr> update(const farr &h0, const farr &pdotdes) {
    float u1[4], w0[6];
    wlConUpdate(&wl, u1, w0, vec(h0, 6, "h0"), vec(pdotdes, 6, "pdotdes"));
    return std::make_tuple(out(u1, 4), out(w0, 6));
  }
};

}  // namespace

PYBIND11_MODULE(_uprightmpc2py, m) {
  m.doc() = "compiled binding of libumpc_mi355x.so's umpcInit / umpcUpdate / wlConInit / wlConUpdate (uprightmpc2py.cpp:30-80)";
  py::class_<UprightMPC2>(m, "UprightMPC2C")
      .def(py::init<float, float, float, float, float, float, float, float, float, float, float, const farr &, int>(), py::arg("dt"),
           py::arg("g"), py::arg("TtoWmax"), py::arg("ws"), py::arg("wds"), py::arg("wpr"), py::arg("wpf"), py::arg("wvr"), py::arg("wvf"),
           py::arg("wthrust"), py::arg("wmom"), py::arg("Ib"), py::arg("maxIter"))
      .def("update", &UprightMPC2::update, py::arg("p0"), py::arg("R0"), py::arg("dq0"), py::arg("pdes"), py::arg("dpdes"),
           py::arg("sdes"), py::arg("actualT0") = -1.0f)
      .def("vectors", &UprightMPC2::vectors)
      .def("matrices", &UprightMPC2::matrices)
      .def("status", &UprightMPC2::status)
      .def("set_compat", &UprightMPC2::set_compat)
      .def_property("T0", &UprightMPC2::T0, &UprightMPC2::set_T0);
  py::class_<WLCon>(m, "WLCon")
      .def(py::init<const farr &, const farr &, const farr &, const farr &, const farr &, float, const farr &>(), py::arg("u0"),
           py::arg("umin"), py::arg("umax"), py::arg("dumax"), py::arg("Qw"), py::arg("controlRate"), py::arg("popts"))
      .def("update", &WLCon::update, py::arg("h0"), py::arg("pdotdes"));
}
