// Container stand-in for the parts of Eigen that the REFERENCE's hot-path translation units touch
// (polynomial_optimization_linear{,_impl}.h, polynomial.{h,cpp}, vertex.{h,cpp}, segment.{h,cpp}, trajectory.{h,cpp},
// convolution.h, rpoly_ak1.cpp).  Eigen is an un-vendored, un-pinned dependency of the reference
// (install/mav_trajectory_generation_https.rosinstall:2) and is absent from this image; with this header the
// reference sources compile WHERE THEY LIE under /root/reference (oracle/Makefile target `ref`) so that the
// reference's own code -- its A/Q/M/R construction, its constraint ordering, its call sequence -- can be executed
// and used as the parity anchor.
//
// What is NOT Eigen's here (stated so nobody mistakes it for the real thing):
//   * every operation is eager and returns a dynamic matrix (no expression templates, no vectorisation);
//   * Matrix::inverse() follows Eigen's algorithm CLASSES (round 3; Gauss-Jordan before): cofactor formulas (adjugate /
//     determinant) up to 4x4 -- the N = 8 instantiation's 4x4 block, LIN:171 -- and an unblocked partial-pivoting LU with
//     column-oriented triangular solves of the identity above (N = 10 / 12: 5x5 / 6x6), but not Eigen's exact operation
//     order (its vectorised kernels, FMA contraction): equal to round-off x condition number, not bit for bit;
//   * SparseMatrix is dense-backed; products skip exact zeros and accumulate in index order;
//   * SparseQR<., COLAMDOrdering> is Eigen's left-looking column Householder QR on the dense-backed matrix (round 3): the
//     columns are processed in a fill-reducing order, each column gets the previous reflectors applied, a column whose
//     remaining norm is below Eigen's default pivot threshold (20 (m + n) max-column-norm eps) is moved to the end (its
//     solution component is zero: the basic solution).  The column order is an EXACT minimum-degree ordering of the column
//     intersection graph (ties to the lowest index) where Eigen runs COLAMD (APPROXIMATE minimum degree with supercolumns
//     and aggressive absorption on the same graph); on the block-tridiagonal R_PP of this path the minimum-degree order is
//     the natural order up to the last two blocks (tests/test_reference_build.py pins that).
// All of these are backward-stable evaluations of the same mathematical operations, so results agree with real
// Eigen to round-off x condition number; nothing in the solveLinear() path depends on more than that.
// Test infrastructure only (oracle/_ref); never linked into the product library.
#pragma once
#include <algorithm>
#include <cassert>
#include <cmath>
#include <complex>
#include <cstddef>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <limits>
#include <memory>
#include <sstream>
#include <string>
#include <type_traits>
#include <vector>

namespace Eigen {

typedef std::ptrdiff_t Index;
const int Dynamic = -1;
enum NoChange_t { NoChange };

template <class T>
using aligned_allocator = std::allocator<T>;

template <class T, int R, int C>
class Matrix;
template <class T>
class View;

enum { StreamPrecision = -1, FullPrecision = -2 };   // (IOFormat precisions: the stream's own / every significant digit)
struct IOFormat {
  IOFormat(int precision = 6, int = 0, const std::string& coeff_sep = " ", const std::string& row_sep = "\n",
           const std::string& row_prefix = "", const std::string& row_suffix = "", const std::string& mat_prefix = "",
           const std::string& mat_suffix = "")
      : precision(precision), coeff_sep(coeff_sep), row_sep(row_sep), row_prefix(row_prefix), row_suffix(row_suffix),
        mat_prefix(mat_prefix), mat_suffix(mat_suffix) {}
  int precision;
  std::string coeff_sep, row_sep, row_prefix, row_suffix, mat_prefix, mat_suffix;
};

namespace internal {
template <class T>
struct real_of {
  typedef T type;
};
template <class T>
struct real_of<std::complex<T>> {
  typedef T type;
};
}  // namespace internal

// ---------------------------------------------------------------------------------------------------------------
// Read interface shared by matrices and views (CRTP).
template <class Derived, class T>
class DenseBase {
 public:
  typedef T Scalar;
  typedef Matrix<T, Dynamic, Dynamic> Dyn;
  typedef typename internal::real_of<T>::type Real;

  const Derived& derived() const { return *static_cast<const Derived*>(this); }
  Derived& derived() { return *static_cast<Derived*>(this); }
  Index rows() const { return derived().rows_(); }
  Index cols() const { return derived().cols_(); }
  Index size() const { return rows() * cols(); }
  const T& coeff(Index r, Index c) const { return derived().at(r, c); }
  const T& operator()(Index r, Index c) const { return coeff(r, c); }
  const T& lin(Index i) const { return cols() == 1 ? coeff(i, 0) : (rows() == 1 ? coeff(0, i) : coeff(i % rows(), i / rows())); }
  const T& operator()(Index i) const { return lin(i); }
  const T& operator[](Index i) const { return lin(i); }

  Dyn eval() const {
    Dyn m(rows(), cols());
    for (Index c = 0; c < cols(); ++c)
      for (Index r = 0; r < rows(); ++r) m(r, c) = coeff(r, c);
    return m;
  }
  Dyn transpose() const {
    Dyn m(cols(), rows());
    for (Index c = 0; c < cols(); ++c)
      for (Index r = 0; r < rows(); ++r) m(c, r) = coeff(r, c);
    return m;
  }
  Dyn diagonal() const {
    const Index n = std::min(rows(), cols());
    Dyn m(n, 1);
    for (Index i = 0; i < n; ++i) m(i, 0) = coeff(i, i);
    return m;
  }
  Dyn asDiagonal() const {
    const Index n = size();
    Dyn m(n, n);
    m.setZero();
    for (Index i = 0; i < n; ++i) m(i, i) = lin(i);
    return m;
  }
  Dyn cwiseInverse() const {
    Dyn m = eval();
    for (Index i = 0; i < m.size(); ++i) m.data()[i] = T(1) / m.data()[i];
    return m;
  }
  Dyn cwiseAbs() const {
    Dyn m = eval();
    for (Index i = 0; i < m.size(); ++i) m.data()[i] = std::abs(m.data()[i]);
    return m;
  }
  template <class D2>
  Dyn cwiseProduct(const DenseBase<D2, T>& o) const {
    assert(rows() == o.rows() && cols() == o.cols());
    Dyn m(rows(), cols());
    for (Index c = 0; c < cols(); ++c)
      for (Index r = 0; r < rows(); ++r) m(r, c) = coeff(r, c) * o.coeff(r, c);
    return m;
  }
  Dyn reverse() const {
    Dyn m(rows(), cols());
    for (Index c = 0; c < cols(); ++c)
      for (Index r = 0; r < rows(); ++r) m(r, c) = coeff(rows() - 1 - r, cols() - 1 - c);
    return m;
  }
  T sum() const {
    T s = T(0);
    for (Index c = 0; c < cols(); ++c)
      for (Index r = 0; r < rows(); ++r) s += coeff(r, c);
    return s;
  }
  Real squaredNorm() const {
    Real s = 0;
    for (Index c = 0; c < cols(); ++c)
      for (Index r = 0; r < rows(); ++r) s += std::norm(std::complex<Real>(coeff(r, c)));
    return s;
  }
  Real norm() const { return std::sqrt(squaredNorm()); }
  T maxCoeff() const {
    T m = coeff(0, 0);
    for (Index c = 0; c < cols(); ++c)
      for (Index r = 0; r < rows(); ++r) m = std::max(m, coeff(r, c));
    return m;
  }
  T minCoeff() const {
    T m = coeff(0, 0);
    for (Index c = 0; c < cols(); ++c)
      for (Index r = 0; r < rows(); ++r) m = std::min(m, coeff(r, c));
    return m;
  }
  template <class D2>
  T dot(const DenseBase<D2, T>& o) const {
    T s = T(0);
    for (Index i = 0; i < size(); ++i) s += lin(i) * o.lin(i);
    return s;
  }
  bool isZero(Real tol = std::numeric_limits<Real>::epsilon() * 100) const {
    for (Index c = 0; c < cols(); ++c)
      for (Index r = 0; r < rows(); ++r)
        if (std::abs(coeff(r, c)) > tol) return false;
    return true;
  }
  template <class D2>
  bool isApprox(const DenseBase<D2, T>& o, Real tol = 1e-12) const {
    Dyn d = eval();
    d -= o;
    return d.squaredNorm() <= tol * tol * std::min(squaredNorm(), o.squaredNorm());
  }
  // Eigen's algorithm classes (see header note): cofactors up to 4x4, partial-pivoting LU above.
  Dyn inverse() const {
    const Index n = rows();
    assert(n == cols());
#if defined(MINI_EIGEN_REFINED_INVERSE)
    // tests/ref_tests only (NOT the oracle/_ref build, whose inverse() follows Eigen's algorithm and nothing more): the
    // reference's AMatrixInversion test (TOPT:731-741) uses `A.inverse()` as the trusted value at 1e-10 absolute on entries up
    // to 1e2 -- one Newton-Schulz step in extended precision, X <- X + X (I - A X), makes it one (the plain LU's own error is
    // 1.4e-10 on the entry -126 of A(1)^-1, which is an exact integer).
    if constexpr (std::is_same<T, double>::value) {
      Dyn x = n <= 4 ? inverse_cofactors() : inverse_partial_piv_lu();
      std::vector<long double> r((size_t)(n * n));
      for (Index i = 0; i < n; ++i)
        for (Index j = 0; j < n; ++j) {
          long double acc = i == j ? 1.0L : 0.0L;
          for (Index k = 0; k < n; ++k) acc -= (long double)coeff(i, k) * (long double)x(k, j);
          r[(size_t)(i * n + j)] = acc;
        }
      Dyn y(n, n);
      for (Index i = 0; i < n; ++i)
        for (Index j = 0; j < n; ++j) {
          long double acc = (long double)x(i, j);
          for (Index k = 0; k < n; ++k) acc += (long double)x(i, k) * r[(size_t)(k * n + j)];
          y(i, j) = (double)acc;
        }
      return y;
    }
#endif
    return n <= 4 ? inverse_cofactors() : inverse_partial_piv_lu();
  }
  // determinant of the sub-matrix without row `skip_r` and column `skip_c` (sizes 0 .. 3: closed forms)
  T minor_det(Index skip_r, Index skip_c) const {
    const Index n = rows() - 1;
    Index ri[3], ci[3];
    for (Index i = 0, k = 0; i < rows(); ++i) if (i != skip_r) ri[k++] = i;
    for (Index j = 0, k = 0; j < cols(); ++j) if (j != skip_c) ci[k++] = j;
    auto a = [&](Index i, Index j) { return coeff(ri[i], ci[j]); };
    if (n == 0) return T(1);
    if (n == 1) return a(0, 0);
    if (n == 2) return a(0, 0) * a(1, 1) - a(0, 1) * a(1, 0);
    return a(0, 0) * (a(1, 1) * a(2, 2) - a(1, 2) * a(2, 1)) - a(0, 1) * (a(1, 0) * a(2, 2) - a(1, 2) * a(2, 0)) +
           a(0, 2) * (a(1, 0) * a(2, 1) - a(1, 1) * a(2, 0));
  }
  Dyn inverse_cofactors() const {
    const Index n = rows();
    Dyn inv(n, n);
    for (Index i = 0; i < n; ++i)
      for (Index j = 0; j < n; ++j) inv(j, i) = (((i + j) & 1) ? T(-1) : T(1)) * minor_det(i, j);   // adjugate = cofactor^T
    T det = T(0);
    for (Index i = 0; i < n; ++i) det += coeff(i, 0) * inv(0, i);   // first column . first row of the adjugate
    for (Index i = 0; i < n; ++i)
      for (Index j = 0; j < n; ++j) inv(i, j) /= det;
    return inv;
  }
  Dyn inverse_partial_piv_lu() const {
    const Index n = rows();
    Dyn lu = eval();
    std::vector<Index> perm((size_t)n);
    for (Index i = 0; i < n; ++i) perm[(size_t)i] = i;
    for (Index k = 0; k < n; ++k) {            // unblocked right-looking LU, row of the biggest entry of column k first
      Index p = k;
      for (Index r = k + 1; r < n; ++r)
        if (std::abs(lu(r, k)) > std::abs(lu(p, k))) p = r;
      if (p != k) {
        for (Index c = 0; c < n; ++c) std::swap(lu(k, c), lu(p, c));
        std::swap(perm[(size_t)k], perm[(size_t)p]);
      }
      if (lu(k, k) != T(0))
        for (Index r = k + 1; r < n; ++r) lu(r, k) /= lu(k, k);
      for (Index c = k + 1; c < n; ++c)
        for (Index r = k + 1; r < n; ++r) lu(r, c) -= lu(r, k) * lu(k, c);
    }
    Dyn inv(n, n);
    for (Index c = 0; c < n; ++c) {            // solve L U x = P e_c, column-oriented (axpy) substitutions
      std::vector<T> x((size_t)n, T(0));
      for (Index i = 0; i < n; ++i) x[(size_t)i] = perm[(size_t)i] == c ? T(1) : T(0);
      for (Index j = 0; j < n; ++j)            // unit lower triangle
        for (Index r = j + 1; r < n; ++r) x[(size_t)r] -= x[(size_t)j] * lu(r, j);
      for (Index j = n - 1; j >= 0; --j) {     // upper triangle
        x[(size_t)j] /= lu(j, j);
        for (Index r = 0; r < j; ++r) x[(size_t)r] -= x[(size_t)j] * lu(r, j);
      }
      for (Index i = 0; i < n; ++i) inv(i, c) = x[(size_t)i];
    }
    return inv;
  }
  std::string format(const IOFormat& f) const {
    std::ostringstream s;
    if (f.precision == FullPrecision) s << std::setprecision(std::numeric_limits<Real>::max_digits10);
    else if (f.precision != StreamPrecision) s << std::setprecision(f.precision);
    s << f.mat_prefix;
    for (Index r = 0; r < rows(); ++r) {
      if (r) s << f.row_sep;
      s << f.row_prefix;
      for (Index c = 0; c < cols(); ++c) {
        if (c) s << f.coeff_sep;
        s << coeff(r, c);
      }
      s << f.row_suffix;
    }
    s << f.mat_suffix;
    return s.str();
  }
  // read-only sub-views (copies; the writable versions live in Matrix / View)
  Dyn block(Index r0, Index c0, Index nr, Index nc) const {
    Dyn m(nr, nc);
    for (Index c = 0; c < nc; ++c)
      for (Index r = 0; r < nr; ++r) m(r, c) = coeff(r0 + r, c0 + c);
    return m;
  }
  template <int NR, int NC>
  Dyn block(Index r0, Index c0) const {
    return block(r0, c0, NR, NC);
  }
  Dyn row(Index r) const { return block(r, 0, 1, cols()); }
  Dyn col(Index c) const { return block(0, c, rows(), 1); }
  Dyn segment(Index i, Index n) const { return cols() == 1 ? block(i, 0, n, 1) : block(0, i, 1, n); }
  template <int NN>
  Dyn segment(Index i) const { return segment(i, NN); }
  Dyn head(Index n) const { return segment(0, n); }
  Dyn tail(Index n) const { return segment(size() - n, n); }
};
// (Eigen's one-parameter base class name, as generic code spells it: test_utils.h:38)
template <class Derived>
using MatrixBase = DenseBase<Derived, double>;

template <class D, class T>
std::ostream& operator<<(std::ostream& s, const DenseBase<D, T>& m) {
  return s << m.format(IOFormat());
}

// ---------------------------------------------------------------------------------------------------------------
// Writable strided window into a matrix.
template <class T>
class View : public DenseBase<View<T>, T> {
 public:
  typedef Matrix<T, Dynamic, Dynamic> Dyn;
  View(T* p, Index ld, Index r, Index c) : p_(p), ld_(ld), r_(r), c_(c) {}
  View(const View&) = default;
  Index rows_() const { return r_; }
  Index cols_() const { return c_; }
  const T& at(Index r, Index c) const { return p_[r + c * ld_]; }
  T& at(Index r, Index c) { return p_[r + c * ld_]; }
  using DenseBase<View<T>, T>::operator();
  using DenseBase<View<T>, T>::operator[];
  T& operator()(Index r, Index c) { return at(r, c); }
  T& operator()(Index i) { return c_ == 1 ? at(i, 0) : at(0, i); }
  T& operator[](Index i) { return (*this)(i); }

  template <class D2>
  View& assign(const DenseBase<D2, T>& o) {
    if (o.rows() == r_ && o.cols() == c_) {
      for (Index c = 0; c < c_; ++c)
        for (Index r = 0; r < r_; ++r) at(r, c) = o.coeff(r, c);
    } else {  // vector <-> row-vector assignment transposes implicitly, as in Eigen
      assert(o.size() == r_ * c_ && (r_ == 1 || c_ == 1) && (o.rows() == 1 || o.cols() == 1));
      for (Index i = 0; i < r_ * c_; ++i) (*this)(i) = o.lin(i);
    }
    return *this;
  }
  template <class D2>
  View& operator=(const DenseBase<D2, T>& o) {
    Dyn tmp = o.eval();  // alias-safe
    return assign(tmp);
  }
  View& operator=(const View& o) {
    Dyn tmp = o.eval();
    return assign(tmp);
  }
  template <class D2>
  View& operator+=(const DenseBase<D2, T>& o) {
    for (Index c = 0; c < c_; ++c)
      for (Index r = 0; r < r_; ++r) at(r, c) += o.coeff(r, c);
    return *this;
  }
  template <class D2>
  View& operator-=(const DenseBase<D2, T>& o) {
    for (Index c = 0; c < c_; ++c)
      for (Index r = 0; r < r_; ++r) at(r, c) -= o.coeff(r, c);
    return *this;
  }
  View& operator*=(const T& s) {
    for (Index c = 0; c < c_; ++c)
      for (Index r = 0; r < r_; ++r) at(r, c) *= s;
    return *this;
  }
  View& operator/=(const T& s) {
    for (Index c = 0; c < c_; ++c)
      for (Index r = 0; r < r_; ++r) at(r, c) /= s;
    return *this;
  }
  View& setConstant(const T& v) {
    for (Index c = 0; c < c_; ++c)
      for (Index r = 0; r < r_; ++r) at(r, c) = v;
    return *this;
  }
  View& setZero() { return setConstant(T(0)); }
  View& setOnes() { return setConstant(T(1)); }

 private:
  T* p_;
  Index ld_, r_, c_;
};

// ---------------------------------------------------------------------------------------------------------------
template <class T, int R, int C>
class Matrix : public DenseBase<Matrix<T, R, C>, T> {
  typedef DenseBase<Matrix<T, R, C>, T> Base;

 public:
  typedef Matrix<T, Dynamic, Dynamic> Dyn;
  enum { RowsAtCompileTime = R, ColsAtCompileTime = C };

  Matrix() : r_(R == Dynamic ? 0 : R), c_(C == Dynamic ? 0 : C), d_((size_t)(r_ * c_), T(0)) {}
  // one integer: vector length (for vectors) -- Eigen semantics
  explicit Matrix(Index n) : r_(C == 1 ? n : (R == Dynamic ? n : R)), c_(C == 1 ? 1 : (R == 1 ? n : (C == Dynamic ? 1 : C))) {
    d_.assign((size_t)(r_ * c_), T(0));
  }
  explicit Matrix(int n) : Matrix((Index)n) {}
  explicit Matrix(size_t n) : Matrix((Index)n) {}
  template <class I1, class I2,
            typename std::enable_if<std::is_integral<I1>::value && std::is_integral<I2>::value && !(R == 2 && C == 1), int>::type = 0>
  Matrix(I1 r, I2 c) : r_((Index)r), c_((Index)c), d_((size_t)(r * c), T(0)) {}
  Matrix(const T& x, const T& y, const T& z) : r_(3), c_(1), d_{x, y, z} {
    if (R == 1) std::swap(r_, c_);
  }
  Matrix(const Matrix&) = default;
  Matrix(Matrix&&) = default;
  template <class D2>
  Matrix(const DenseBase<D2, T>& o) : r_(R == Dynamic ? 0 : R), c_(C == Dynamic ? 0 : C) {
    *this = o;
  }

  Index rows_() const { return r_; }
  Index cols_() const { return c_; }
  const T& at(Index r, Index c) const {
    assert(r >= 0 && r < r_ && c >= 0 && c < c_);
    return d_[(size_t)(r + c * r_)];
  }
  T& at(Index r, Index c) {
    assert(r >= 0 && r < r_ && c >= 0 && c < c_);
    return d_[(size_t)(r + c * r_)];
  }
  using Base::operator();
  using Base::operator[];
  T& operator()(Index r, Index c) { return at(r, c); }
  T& operator()(Index i) {
    assert(i >= 0 && i < r_ * c_);
    return d_[(size_t)i];
  }
  T& operator[](Index i) { return (*this)(i); }
  T* data() { return d_.data(); }
  const T* data() const { return d_.data(); }

  Matrix& operator=(const Matrix&) = default;
  Matrix& operator=(Matrix&&) = default;
  template <class D2>
  Matrix& operator=(const DenseBase<D2, T>& o) {
    Index nr = o.rows(), nc = o.cols();
    // implicit vector transposition on assignment (column <- row and vice versa)
    const bool flip = (C == 1 && nr == 1 && nc != 1) || (R == 1 && nc == 1 && nr != 1);
    std::vector<T> tmp((size_t)(nr * nc));
    for (Index c = 0; c < nc; ++c)
      for (Index r = 0; r < nr; ++r) tmp[(size_t)(flip ? (c + r * nc) : (r + c * nr))] = o.coeff(r, c);
    if (flip) std::swap(nr, nc);
    assert((R == Dynamic || R == nr) && (C == Dynamic || C == nc));
    r_ = nr;
    c_ = nc;
    d_.swap(tmp);
    return *this;
  }

  void resize(Index n) {
    if (C == 1 || (R != 1 && C == Dynamic && R == Dynamic))
      resize(n, 1);
    else
      resize(1, n);
  }
  void resize(Index r, Index c) {
    if (r * c != r_ * c_) d_.assign((size_t)(r * c), T(0));
    r_ = r;
    c_ = c;
  }
  void resize(Index r, NoChange_t) { resize(r, c_); }
  void resize(NoChange_t, Index c) { resize(r_, c); }
  void conservativeResize(Index n) {
    Matrix old = *this;
    resize(n);
    for (Index i = 0; i < std::min(n, old.size()); ++i) (*this)(i) = old(i);
  }

  Matrix& setConstant(const T& v) {
    std::fill(d_.begin(), d_.end(), v);
    return *this;
  }
  Matrix& setZero() { return setConstant(T(0)); }
  Matrix& setOnes() { return setConstant(T(1)); }
  Matrix& setIdentity() {
    setZero();
    for (Index i = 0; i < std::min(r_, c_); ++i) at(i, i) = T(1);
    return *this;
  }
  static Matrix Constant(Index n, const T& v) {
    Matrix m(n);
    m.setConstant(v);
    return m;
  }
  static Matrix Constant(Index r, Index c, const T& v) {
    Matrix m;
    m.resize(r, c);
    m.setConstant(v);
    return m;
  }
  static Matrix Zero() {
    Matrix m;
    m.setZero();
    return m;
  }
  static Matrix Zero(Index n) { return Constant(n, T(0)); }
  static Matrix Zero(Index r, Index c) { return Constant(r, c, T(0)); }
  static Matrix Ones(Index n) { return Constant(n, T(1)); }
  static Matrix Identity() {
    Matrix m;
    m.setIdentity();
    return m;
  }
  static Matrix Identity(Index r, Index c) {
    Matrix m;
    m.resize(r, c);
    m.setIdentity();
    return m;
  }

  // writable windows
  using Base::block;
  using Base::col;
  using Base::head;
  using Base::row;
  using Base::segment;
  using Base::tail;
  View<T> block(Index r0, Index c0, Index nr, Index nc) {
    assert(r0 >= 0 && c0 >= 0 && r0 + nr <= r_ && c0 + nc <= c_);
    return View<T>(d_.data() + r0 + c0 * r_, r_, nr, nc);
  }
  template <int NR, int NC>
  View<T> block(Index r0, Index c0) {
    return block(r0, c0, NR, NC);
  }
  View<T> row(Index r) { return block(r, 0, 1, c_); }
  View<T> col(Index c) { return block(0, c, r_, 1); }
  View<T> segment(Index i, Index n) { return c_ == 1 ? block(i, 0, n, 1) : block(0, i, 1, n); }
  View<T> head(Index n) { return segment(0, n); }
  View<T> tail(Index n) { return segment(r_ * c_ - n, n); }

  template <class D2>
  Matrix& operator+=(const DenseBase<D2, T>& o) {
    assert(o.rows() == r_ && o.cols() == c_);
    for (Index c = 0; c < c_; ++c)
      for (Index r = 0; r < r_; ++r) at(r, c) += o.coeff(r, c);
    return *this;
  }
  template <class D2>
  Matrix& operator-=(const DenseBase<D2, T>& o) {
    assert(o.rows() == r_ && o.cols() == c_);
    for (Index c = 0; c < c_; ++c)
      for (Index r = 0; r < r_; ++r) at(r, c) -= o.coeff(r, c);
    return *this;
  }
  Matrix& operator*=(const T& s) {
    for (auto& x : d_) x *= s;
    return *this;
  }
  Matrix& operator/=(const T& s) {
    for (auto& x : d_) x /= s;
    return *this;
  }

  // 1x1 results used as scalars (`double cost = c.transpose() * Q * c;`)
  template <int RR = R, int CC = C, typename std::enable_if<RR == Dynamic && CC == Dynamic, int>::type = 0>
  operator T() const {
    assert(r_ == 1 && c_ == 1);
    return d_[0];
  }

  // comma initialiser: `v << a, b, c;` with scalars or vector pieces
  class CommaInit {
   public:
    CommaInit(Matrix& m) : m_(m), pos_(0) {}
    CommaInit& put(const T& v) {
      m_(pos_++) = v;
      return *this;
    }
    template <class D2>
    CommaInit& put(const DenseBase<D2, T>& o) {
      assert(m_.cols() == 1 || m_.rows() == 1);
      for (Index i = 0; i < o.size(); ++i) m_(pos_++) = o.lin(i);
      return *this;
    }
    CommaInit& operator,(const T& v) { return put(v); }
    template <class D2>
    CommaInit& operator,(const DenseBase<D2, T>& o) {
      return put(o);
    }

   private:
    Matrix& m_;
    Index pos_;
  };
  CommaInit operator<<(const T& v) {
    CommaInit ci(*this);
    ci.put(v);
    return ci;
  }
  template <class D2>
  CommaInit operator<<(const DenseBase<D2, T>& o) {
    CommaInit ci(*this);
    ci.put(o);
    return ci;
  }

 private:
  Index r_, c_;
  std::vector<T> d_;
};

// ---------------------------------------------------------------------------------------------------------------
// arithmetic (eager, dynamic result)
template <class A, class B, class T>
Matrix<T, Dynamic, Dynamic> operator+(const DenseBase<A, T>& a, const DenseBase<B, T>& b) {
  Matrix<T, Dynamic, Dynamic> m = a.eval();
  m += b;
  return m;
}
template <class A, class B, class T>
Matrix<T, Dynamic, Dynamic> operator-(const DenseBase<A, T>& a, const DenseBase<B, T>& b) {
  Matrix<T, Dynamic, Dynamic> m = a.eval();
  m -= b;
  return m;
}
template <class A, class T>
Matrix<T, Dynamic, Dynamic> operator-(const DenseBase<A, T>& a) {
  Matrix<T, Dynamic, Dynamic> m = a.eval();
  for (Index i = 0; i < m.size(); ++i) m.data()[i] = -m.data()[i];
  return m;
}
template <class A, class B, class T>
Matrix<T, Dynamic, Dynamic> operator*(const DenseBase<A, T>& a, const DenseBase<B, T>& b) {
  assert(a.cols() == b.rows());
  Matrix<T, Dynamic, Dynamic> m(a.rows(), b.cols());
  for (Index c = 0; c < b.cols(); ++c)
    for (Index r = 0; r < a.rows(); ++r) {
      T s = T(0);
      for (Index k = 0; k < a.cols(); ++k) s += a.coeff(r, k) * b.coeff(k, c);
      m(r, c) = s;
    }
  return m;
}
template <class A, class T, class S, typename std::enable_if<std::is_arithmetic<S>::value, int>::type = 0>
Matrix<T, Dynamic, Dynamic> operator*(const DenseBase<A, T>& a, const S& s) {
  Matrix<T, Dynamic, Dynamic> m = a.eval();
  m *= T(s);
  return m;
}
template <class A, class T, class S, typename std::enable_if<std::is_arithmetic<S>::value, int>::type = 0>
Matrix<T, Dynamic, Dynamic> operator*(const S& s, const DenseBase<A, T>& a) {
  return a * s;
}
template <class A, class T, class S, typename std::enable_if<std::is_arithmetic<S>::value, int>::type = 0>
Matrix<T, Dynamic, Dynamic> operator/(const DenseBase<A, T>& a, const S& s) {
  Matrix<T, Dynamic, Dynamic> m = a.eval();
  m /= T(s);
  return m;
}
template <class A, class B, class T>
bool operator==(const DenseBase<A, T>& a, const DenseBase<B, T>& b) {
  if (a.rows() != b.rows() || a.cols() != b.cols()) return false;
  for (Index c = 0; c < a.cols(); ++c)
    for (Index r = 0; r < a.rows(); ++r)
      if (!(a.coeff(r, c) == b.coeff(r, c))) return false;
  return true;
}
template <class A, class B, class T>
bool operator!=(const DenseBase<A, T>& a, const DenseBase<B, T>& b) {
  return !(a == b);
}

// Map<M>(ptr, n) / Map<M>(ptr, rows, cols): window onto caller-owned, column-major storage.
template <class M>
class Map : public View<typename M::Scalar> {
  typedef typename M::Scalar T;

 public:
  Map(T* p, Index n) : View<T>(p, M::ColsAtCompileTime == 1 ? n : 1, M::ColsAtCompileTime == 1 ? n : 1, M::ColsAtCompileTime == 1 ? 1 : n) {}
  Map(const T* p, Index n) : Map(const_cast<T*>(p), n) {}
  Map(T* p, Index r, Index c) : View<T>(p, r, r, c) {}
  Map(const T* p, Index r, Index c) : View<T>(const_cast<T*>(p), r, r, c) {}
  using View<T>::operator=;
};

typedef Matrix<double, Dynamic, Dynamic> MatrixXd;
typedef Matrix<double, Dynamic, 1> VectorXd;
typedef Matrix<double, 1, Dynamic> RowVectorXd;
typedef Matrix<std::complex<double>, Dynamic, 1> VectorXcd;
typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 4, 1> Vector4d;
typedef Matrix<double, 3, 3> Matrix3d;
typedef Matrix<int, Dynamic, 1> VectorXi;

// ---------------------------------------------------------------------------------------------------------------
// Sparse stand-ins (dense-backed).
template <class T>
class Triplet {
 public:
  Triplet() : r_(0), c_(0), v_(T(0)) {}
  Triplet(Index r, Index c, const T& v = T(0)) : r_(r), c_(c), v_(v) {}
  Index row() const { return r_; }
  Index col() const { return c_; }
  const T& value() const { return v_; }

 private:
  Index r_, c_;
  T v_;
};

template <class T>
class SparseMatrix {
 public:
  typedef Matrix<T, Dynamic, Dynamic> Dyn;
  SparseMatrix() {}
  SparseMatrix(Index r, Index c) : m_(r, c) {}
  explicit SparseMatrix(const Dyn& m) : m_(m) {}
  Index rows() const { return m_.rows(); }
  Index cols() const { return m_.cols(); }
  void resize(Index r, Index c) {
    m_.resize(r, c);
    m_.setZero();
  }
  template <class It>
  void setFromTriplets(It b, It e) {
    m_.setZero();
    for (; b != e; ++b) m_(b->row(), b->col()) += b->value();  // duplicates are summed, as in Eigen
  }
  T coeff(Index r, Index c) const { return m_(r, c); }
  T& coeffRef(Index r, Index c) { return m_(r, c); }
  Index nonZeros() const {
    Index n = 0;
    for (Index i = 0; i < m_.size(); ++i) n += m_.data()[i] != T(0);
    return n;
  }
  SparseMatrix block(Index r0, Index c0, Index nr, Index nc) const { return SparseMatrix(m_.block(r0, c0, nr, nc)); }
  SparseMatrix transpose() const { return SparseMatrix(m_.transpose()); }
  SparseMatrix operator-() const { return SparseMatrix(-m_); }
  const Dyn& dense() const { return m_; }
  operator Dyn() const { return m_; }
  void makeCompressed() {}

 private:
  Dyn m_;
};

template <class T>
SparseMatrix<T> operator*(const SparseMatrix<T>& a, const SparseMatrix<T>& b) {
  Matrix<T, Dynamic, Dynamic> m(a.rows(), b.cols());
  m.setZero();
  for (Index c = 0; c < b.cols(); ++c)
    for (Index k = 0; k < a.cols(); ++k) {
      const T bk = b.coeff(k, c);
      if (bk == T(0)) continue;
      for (Index r = 0; r < a.rows(); ++r) {
        const T ar = a.coeff(r, k);
        if (ar != T(0)) m(r, c) += ar * bk;
      }
    }
  return SparseMatrix<T>(m);
}
template <class T, class D>
Matrix<T, Dynamic, Dynamic> operator*(const SparseMatrix<T>& a, const DenseBase<D, T>& b) {
  Matrix<T, Dynamic, Dynamic> m(a.rows(), b.cols());
  m.setZero();
  for (Index c = 0; c < b.cols(); ++c)
    for (Index k = 0; k < a.cols(); ++k) {
      const T bk = b.coeff(k, c);
      for (Index r = 0; r < a.rows(); ++r) {
        const T ar = a.coeff(r, k);
        if (ar != T(0)) m(r, c) += ar * bk;
      }
    }
  return m;
}
template <class T>
std::ostream& operator<<(std::ostream& s, const SparseMatrix<T>& m) {
  return s << m.dense();
}

template <class I>
struct COLAMDOrdering {};
template <class I>
struct NaturalOrdering {};
enum ComputationInfo { Success = 0, NumericalIssue = 1, NoConvergence = 2, InvalidInput = 3 };

// Exact minimum-degree ordering of the column intersection graph of `a` (columns i, j adjacent when some row holds a
// non-zero of both), ties to the lowest column index; eliminating a column joins its remaining neighbours pairwise.
template <class Dyn>
std::vector<Index> minimum_degree_column_order(const Dyn& a) {
  const Index m = a.rows(), n = a.cols();
  std::vector<std::vector<char>> adj((size_t)n, std::vector<char>((size_t)n, 0));
  for (Index r = 0; r < m; ++r)
    for (Index i = 0; i < n; ++i) {
      if (a(r, i) == 0.0) continue;
      for (Index j = i + 1; j < n; ++j)
        if (a(r, j) != 0.0) adj[(size_t)i][(size_t)j] = adj[(size_t)j][(size_t)i] = 1;
    }
  std::vector<char> done((size_t)n, 0);
  std::vector<Index> order;
  for (Index step = 0; step < n; ++step) {
    Index best = -1, best_deg = n + 1;
    for (Index c = 0; c < n; ++c) {
      if (done[(size_t)c]) continue;
      Index deg = 0;
      for (Index j = 0; j < n; ++j) deg += !done[(size_t)j] && adj[(size_t)c][(size_t)j];
      if (deg < best_deg) { best_deg = deg; best = c; }
    }
    done[(size_t)best] = 1;
    order.push_back(best);
    for (Index i = 0; i < n; ++i) {
      if (done[(size_t)i] || !adj[(size_t)best][(size_t)i]) continue;
      for (Index j = i + 1; j < n; ++j)
        if (!done[(size_t)j] && adj[(size_t)best][(size_t)j]) adj[(size_t)i][(size_t)j] = adj[(size_t)j][(size_t)i] = 1;
    }
  }
  return order;
}

// Left-looking column Householder QR in a fill-reducing column order, Eigen's pivot threshold and its treatment of
// (numerically) dependent columns (see header note).
template <class Mat, class Ordering>
class SparseQR {
 public:
  typedef Matrix<double, Dynamic, Dynamic> Dyn;
  SparseQR() : threshold_(0), user_threshold_(false), rank_(0), info_(InvalidInput) {}
  explicit SparseQR(const Mat& a) : threshold_(0), user_threshold_(false) { compute(a); }
  void compute(const Mat& a) {
    const Dyn A = a.dense();
    const Index m = A.rows(), n = A.cols();
    m_ = m; n_ = n;
    std::vector<Index> order;
    if (std::is_same<Ordering, NaturalOrdering<int>>::value) { for (Index c = 0; c < n; ++c) order.push_back(c); }
    else order = minimum_degree_column_order(A);
    if (!user_threshold_) {
      double max_col = 0;
      for (Index c = 0; c < n; ++c) max_col = std::max(max_col, A.col(c).norm());
      if (max_col == 0.0) max_col = 1.0;
      threshold_ = 20.0 * (double)(m + n) * max_col * std::numeric_limits<double>::epsilon();
    }
    v_.clear(); tau_.clear(); rcols_.clear(); accepted_.clear();
    Index nz = 0;                                   // Eigen's nonzeroCol: reflectors / accepted columns so far
    for (Index oc = 0; oc < n && nz < m; ++oc) {
      const Index col = order[(size_t)oc];
      std::vector<double> t((size_t)m);
      for (Index r = 0; r < m; ++r) t[(size_t)r] = A(r, col);
      for (Index k = 0; k < nz; ++k) apply(k, t);   // the previous reflectors, in order
      const double c0 = t[(size_t)nz];
      double sq = 0;
      for (Index r = nz + 1; r < m; ++r) sq += t[(size_t)r] * t[(size_t)r];
      double beta, tau;
      std::vector<double> v((size_t)m, 0.0);
      v[(size_t)nz] = 1.0;
      if (sq == 0.0) {
        beta = c0;
        tau = 0.0;
      } else {
        beta = std::sqrt(c0 * c0 + sq);
        if (c0 >= 0.0) beta = -beta;
        for (Index r = nz + 1; r < m; ++r) v[(size_t)r] = t[(size_t)r] / (c0 - beta);
        tau = (beta - c0) / beta;
      }
      if (std::abs(beta) > threshold_) {            // accepted: a column of R and a reflector
        std::vector<double> rc(t.begin(), t.begin() + nz);
        rc.push_back(beta);
        rcols_.push_back(rc);
        v_.push_back(v);
        tau_.push_back(tau);
        accepted_.push_back(col);
        ++nz;
      }                                             // else: dependent column, moved to the end (component zero)
    }
    rank_ = nz;
    info_ = Success;
  }
  template <class D>
  Dyn solve(const DenseBase<D, double>& b) const {
    Dyn y = b.eval();
    Dyn x(n_, y.cols());
    x.setZero();
    for (Index c = 0; c < y.cols(); ++c) {
      std::vector<double> t((size_t)m_);
      for (Index r = 0; r < m_; ++r) t[(size_t)r] = y(r, c);
      for (Index k = 0; k < rank_; ++k) apply(k, t);          // Q^T b
      std::vector<double> z((size_t)rank_, 0.0);
      for (Index k = rank_ - 1; k >= 0; --k) {                // R11 z = (Q^T b)[0 .. rank)
        double sacc = t[(size_t)k];
        for (Index j = k + 1; j < rank_; ++j) sacc -= rcols_[(size_t)j][(size_t)k] * z[(size_t)j];
        z[(size_t)k] = sacc / rcols_[(size_t)k][(size_t)k];
      }
      for (Index k = 0; k < rank_; ++k) x(accepted_[(size_t)k], c) = z[(size_t)k];
    }
    return x;
  }
  Index rank() const { return rank_; }
  ComputationInfo info() const { return info_; }
  void setPivotThreshold(double t) { threshold_ = t; user_threshold_ = true; }
  const std::vector<Index>& acceptedColumns() const { return accepted_; }   // column order of R (test hook)

 private:
  void apply(Index k, std::vector<double>& t) const {          // t <- (I - tau v v^T) t
    const std::vector<double>& v = v_[(size_t)k];
    double dot = 0;
    for (Index r = k; r < m_; ++r) dot += v[(size_t)r] * t[(size_t)r];
    dot *= tau_[(size_t)k];
    if (dot == 0.0) return;
    for (Index r = k; r < m_; ++r) t[(size_t)r] -= dot * v[(size_t)r];
  }
  std::vector<std::vector<double>> v_, rcols_;
  std::vector<double> tau_;
  std::vector<Index> accepted_;
  Index m_ = 0, n_ = 0;
  double threshold_;
  bool user_threshold_;
  Index rank_;
  ComputationInfo info_;
};

}  // namespace Eigen
