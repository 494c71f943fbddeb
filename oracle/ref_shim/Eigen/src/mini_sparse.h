// mini_sparse.h — look-alike of the Eigen/Sparse + SparseCholesky names that the reference's LinearSolverEigen uses
// (TEST INFRASTRUCTURE, see mini_eigen.h).  SimplicialLDLT here = fill-reducing minimum-degree ordering + the up-looking
// LDL^T factorisation (elimination tree, one sparse triangular solve per row) that Eigen's simplicial Cholesky implements;
// like Eigen's LDLT variant it reports NumericalIssue only for an exactly zero pivot.  The ORDERING differs from Eigen's AMD,
// so factors differ from a real Eigen build by floating-point rounding only.
#pragma once
#include <queue>
#include <stdexcept>
#include <utility>
#include "mini_eigen.h"

namespace Eigen {

template <class S, class I = int>
class Triplet {
  I r_, c_; S v_;
 public:
  Triplet() : r_(0), c_(0), v_(0) {}
  Triplet(const I& r, const I& c, const S& v = S(0)) : r_(r), c_(c), v_(v) {}
  const I& row() const { return r_; }
  const I& col() const { return c_; }
  const S& value() const { return v_; }
};

template <int SR, int SC, class I = int>
class PermutationMatrix {
  Matrix<I, Dynamic, 1> idx_;
 public:
  PermutationMatrix() {}
  explicit PermutationMatrix(Index n) { idx_.resize(n); }
  void resize(Index n) { idx_.resize(n); }
  Index size() const { return idx_.size(); }
  Index rows() const { return idx_.size(); }
  Index cols() const { return idx_.size(); }
  Matrix<I, Dynamic, 1>& indices() { return idx_; }
  const Matrix<I, Dynamic, 1>& indices() const { return idx_; }
  void setIdentity(Index n) { idx_.resize(n); for (Index i = 0; i < n; i++) idx_(i) = (I)i; }
  PermutationMatrix inverse() const { PermutationMatrix r(size()); for (Index i = 0; i < size(); i++) r.idx_(idx_(i)) = (I)i; return r; }
  template <int A, int B, class J> PermutationMatrix& operator=(const PermutationMatrix<A, B, J>& o) {
    idx_.resize(o.size()); for (Index i = 0; i < o.size(); i++) idx_(i) = (I)o.indices()(i); return *this;
  }
};

template <class SM, int UpLo> class SelfAdjointViewS;
template <class SM, int UpLo, class P> struct TwistedS { const SM& m; const P& p; };

template <class S, int Opt = ColMajor, class I = int>
class SparseMatrix {
 public:
  typedef S Scalar;
  typedef Eigen::Index Index;
  typedef I StorageIndex;
 private:
  Index rows_ = 0, cols_ = 0;
  std::vector<I> outer_, inner_;
  std::vector<S> val_;
 public:
  SparseMatrix() { outer_.assign(1, 0); }
  SparseMatrix(Index r, Index c) { resize(r, c); }
  void resize(Index r, Index c) { rows_ = r; cols_ = c; outer_.assign((size_t)c + 1, 0); inner_.clear(); val_.clear(); }
  void resizeNonZeros(Index n) { inner_.resize((size_t)n); val_.resize((size_t)n); }
  Index rows() const { return rows_; }
  Index cols() const { return cols_; }
  Index nonZeros() const { return (Index)val_.size(); }
  Index outerSize() const { return cols_; }
  S* valuePtr() { return val_.data(); }
  const S* valuePtr() const { return val_.data(); }
  I* innerIndexPtr() { return inner_.data(); }
  const I* innerIndexPtr() const { return inner_.data(); }
  I* outerIndexPtr() { return outer_.data(); }
  const I* outerIndexPtr() const { return outer_.data(); }
  void makeCompressed() {}
  bool isCompressed() const { return true; }
  // compressed column storage, rows ascending inside a column, duplicates summed
  template <class It> void setFromTriplets(It b, It e) {
    std::vector<std::vector<std::pair<I, S>>> cols((size_t)cols_);
    for (It it = b; it != e; ++it) cols[(size_t)it->col()].push_back({(I)it->row(), it->value()});
    outer_.assign((size_t)cols_ + 1, 0); inner_.clear(); val_.clear();
    for (Index c = 0; c < cols_; c++) {
      auto& v = cols[(size_t)c];
      std::stable_sort(v.begin(), v.end(), [](const std::pair<I, S>& a, const std::pair<I, S>& b2) { return a.first < b2.first; });
      for (size_t k = 0; k < v.size(); k++) {
        if (k && v[k].first == v[k - 1].first) val_.back() += v[k].second;
        else { inner_.push_back(v[k].first); val_.push_back(v[k].second); }
      }
      outer_[(size_t)c + 1] = (I)inner_.size();
    }
  }
  S coeff(Index r, Index c) const {
    for (I p = outer_[(size_t)c]; p < outer_[(size_t)c + 1]; p++) if (inner_[(size_t)p] == r) return val_[(size_t)p];
    return S(0);
  }
  template <int UpLo> SelfAdjointViewS<SparseMatrix, UpLo> selfadjointView() { return SelfAdjointViewS<SparseMatrix, UpLo>(*this); }
  template <int UpLo> SelfAdjointViewS<const SparseMatrix, UpLo> selfadjointView() const { return SelfAdjointViewS<const SparseMatrix, UpLo>(*this); }
  // full symmetric matrix from the stored triangle of a self-adjoint view
  template <class SM2, int UpLo> SparseMatrix& operator=(const SelfAdjointViewS<SM2, UpLo>& v) {
    const auto& a = v.matrix();
    std::vector<Triplet<S, I>> t;
    for (Index c = 0; c < a.cols(); c++)
      for (I p = a.outerIndexPtr()[c]; p < a.outerIndexPtr()[c + 1]; p++) {
        const I r = a.innerIndexPtr()[p];
        if ((UpLo == Upper && r > c) || (UpLo == Lower && r < c)) continue;
        t.push_back(Triplet<S, I>(r, (I)c, a.valuePtr()[p]));
        if (r != c) t.push_back(Triplet<S, I>((I)c, r, a.valuePtr()[p]));
      }
    resize(a.rows(), a.cols());
    setFromTriplets(t.begin(), t.end());
    return *this;
  }
};

template <class SM, int UpLo>
class SelfAdjointViewS {
  SM& m_;
 public:
  explicit SelfAdjointViewS(SM& m) : m_(m) {}
  SM& matrix() const { return m_; }
  template <class P> TwistedS<SM, UpLo, P> twistedBy(const P& p) const { return TwistedS<SM, UpLo, P>{m_, p}; }
  // dest.selfadjointView<DUpLo>() = src.selfadjointView<SUpLo>().twistedBy(P): the DUpLo triangle of P A P^T
  template <class SM2, int SUpLo, class P> SelfAdjointViewS& operator=(const TwistedS<SM2, SUpLo, P>& tw) {
    typedef typename std::remove_const<SM>::type M;
    typedef typename M::Scalar S;
    typedef typename M::StorageIndex I;
    const auto& a = tw.m;
    std::vector<Triplet<S, I>> t;
    for (Index c = 0; c < a.cols(); c++)
      for (I p = a.outerIndexPtr()[c]; p < a.outerIndexPtr()[c + 1]; p++) {
        const I r = a.innerIndexPtr()[p];
        if ((SUpLo == Upper && r > c) || (SUpLo == Lower && r < c)) continue;
        I pr = (I)tw.p.indices()(r), pc = (I)tw.p.indices()(c);
        if ((UpLo == Upper && pr > pc) || (UpLo == Lower && pr < pc)) std::swap(pr, pc);
        t.push_back(Triplet<S, I>(pr, pc, a.valuePtr()[p]));
      }
    m_.resize(a.rows(), a.cols());
    m_.setFromTriplets(t.begin(), t.end());
    return *this;
  }
};

namespace internal {
// Greedy minimum-degree ordering on the elimination graph of a full symmetric pattern (stand-in for Eigen's AMD).
// perm.indices()(k) = the original index eliminated at step k.
template <class SM, class P>
void minimum_degree_ordering(const SM& C, P& perm) {
  typedef typename SM::StorageIndex I;
  const Index n = C.cols();
  std::vector<std::vector<I>> adj((size_t)n);
  for (Index c = 0; c < n; c++)
    for (I p = C.outerIndexPtr()[c]; p < C.outerIndexPtr()[c + 1]; p++) {
      const I r = C.innerIndexPtr()[p];
      if (r != c) { adj[(size_t)c].push_back(r); adj[(size_t)r].push_back((I)c); }
    }
  for (auto& a : adj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
  std::vector<char> done((size_t)n, 0);
  typedef std::pair<size_t, I> QE;   // (degree, node): ties by the smaller index
  std::priority_queue<QE, std::vector<QE>, std::greater<QE>> q;
  for (Index i = 0; i < n; i++) q.push({adj[(size_t)i].size(), (I)i});
  perm.resize(n);
  std::vector<I> merged;
  Index k = 0;
  while (!q.empty()) {
    const QE e = q.top(); q.pop();
    const I v = e.second;
    if (done[(size_t)v] || e.first != adj[(size_t)v].size()) continue;   // stale entry
    done[(size_t)v] = 1;
    perm.indices()(k++) = v;
    const std::vector<I> nb = adj[(size_t)v];
    for (I u : nb) {   // the neighbours become a clique, v leaves the graph
      std::vector<I>& au = adj[(size_t)u];
      merged.clear();
      std::set_union(au.begin(), au.end(), nb.begin(), nb.end(), std::back_inserter(merged));
      au.clear();
      for (I w : merged) if (w != u && w != v) au.push_back(w);
      q.push({au.size(), u});
    }
    std::vector<I>().swap(adj[(size_t)v]);
  }
}
}  // namespace internal

template <class I = int>
struct AMDOrdering {
  template <class SM, class P> void operator()(const SM& C, P& perm) const { internal::minimum_degree_ordering(C, perm); }
};

template <class SM, int UpLo_ = Lower>
class SimplicialLDLT {
 public:
  typedef typename SM::Scalar Scalar;
  typedef typename SM::StorageIndex StorageIndex;
  typedef SM CholMatrixType;
  typedef Matrix<Scalar, Dynamic, 1> VectorType;
  enum { UpLo = UpLo_ };

 protected:
  PermutationMatrix<Dynamic, Dynamic, StorageIndex> m_P, m_Pinv;   // m_Pinv.indices()(k) = original index at permuted position k
  ComputationInfo m_info = Success;
  bool m_analyzed = false;
  Index n_ = 0;
  std::vector<StorageIndex> parent_, lp_, li_, lnz_;
  std::vector<Scalar> lx_, d_;
  SM ap_;   // upper triangle of P A P^T

  // symbolic analysis of the permuted upper triangle: elimination tree and column counts of L
  void analyzePattern_preordered(const SM& ap, bool /*doLDLT*/) {
    const Index n = ap.cols();
    n_ = n;
    parent_.assign((size_t)n, -1); lnz_.assign((size_t)n, 0); lp_.assign((size_t)n + 1, 0);
    std::vector<StorageIndex> flag((size_t)n);
    for (Index k = 0; k < n; k++) {
      flag[(size_t)k] = (StorageIndex)k;
      for (StorageIndex p = ap.outerIndexPtr()[k]; p < ap.outerIndexPtr()[k + 1]; p++) {
        StorageIndex i = ap.innerIndexPtr()[p];
        if (i >= k) continue;
        for (; flag[(size_t)i] != k; i = parent_[(size_t)i]) {
          if (parent_[(size_t)i] == -1) parent_[(size_t)i] = (StorageIndex)k;
          lnz_[(size_t)i]++;
          flag[(size_t)i] = (StorageIndex)k;
        }
      }
    }
    for (Index k = 0; k < n; k++) lp_[(size_t)k + 1] = lp_[(size_t)k] + lnz_[(size_t)k];
    li_.assign((size_t)lp_[(size_t)n], 0); lx_.assign((size_t)lp_[(size_t)n], Scalar(0)); d_.assign((size_t)n, Scalar(0));
    m_analyzed = true;
  }
  void permute_upper(const SM& a) { ap_.template selfadjointView<Upper>() = a.template selfadjointView<UpLo_>().twistedBy(m_P); }

 public:
  SimplicialLDLT() {}
  void analyzePattern(const SM& a) {
    SM C;
    C = a.template selfadjointView<UpLo_>();
    internal::minimum_degree_ordering(C, m_Pinv);
    m_P = m_Pinv.inverse();
    permute_upper(a);
    analyzePattern_preordered(ap_, true);
  }
  void factorize(const SM& a) {
    if (!m_analyzed) throw std::runtime_error("SimplicialLDLT (look-alike): factorize before analyzePattern");
    permute_upper(a);
    const Index n = n_;
    std::vector<Scalar> y((size_t)n, Scalar(0));
    std::vector<StorageIndex> pattern((size_t)n), flag((size_t)n);
    std::fill(lnz_.begin(), lnz_.end(), 0);
    m_info = Success;
    for (Index k = 0; k < n; k++) {
      y[(size_t)k] = Scalar(0);
      Index top = n;
      flag[(size_t)k] = (StorageIndex)k;
      for (StorageIndex p = ap_.outerIndexPtr()[k]; p < ap_.outerIndexPtr()[k + 1]; p++) {
        StorageIndex i = ap_.innerIndexPtr()[p];
        if (i > k) continue;
        y[(size_t)i] += ap_.valuePtr()[p];
        Index len = 0;
        for (; flag[(size_t)i] != k; i = parent_[(size_t)i]) { pattern[(size_t)len++] = i; flag[(size_t)i] = (StorageIndex)k; }
        while (len > 0) pattern[(size_t)--top] = pattern[(size_t)--len];
      }
      Scalar d = y[(size_t)k];
      y[(size_t)k] = Scalar(0);
      for (; top < n; top++) {
        const StorageIndex i = pattern[(size_t)top];
        const Scalar yi = y[(size_t)i];
        y[(size_t)i] = Scalar(0);
        const StorageIndex p2 = lp_[(size_t)i] + lnz_[(size_t)i];
        StorageIndex p;
        for (p = lp_[(size_t)i]; p < p2; p++) y[(size_t)li_[(size_t)p]] -= lx_[(size_t)p] * yi;
        const Scalar l_ki = yi / d_[(size_t)i];
        d -= l_ki * yi;
        li_[(size_t)p] = (StorageIndex)k;
        lx_[(size_t)p] = l_ki;
        lnz_[(size_t)i]++;
      }
      d_[(size_t)k] = d;
      if (d == Scalar(0)) { m_info = NumericalIssue; return; }   // Eigen's LDLT variant: only an exactly zero pivot fails
    }
  }
  void compute(const SM& a) { analyzePattern(a); factorize(a); }
  ComputationInfo info() const { return m_info; }
  template <class B> VectorType solve(const DenseBase<B>& b) const {
    const Index n = n_;
    VectorType x(n);
    for (Index k = 0; k < n; k++) x(k) = b(m_Pinv.indices()(k));                                 // P b
    for (Index j = 0; j < n; j++) for (StorageIndex p = lp_[(size_t)j]; p < lp_[(size_t)j] + lnz_[(size_t)j]; p++) x(li_[(size_t)p]) -= lx_[(size_t)p] * x(j);
    for (Index j = 0; j < n; j++) x(j) /= d_[(size_t)j];
    for (Index j = n - 1; j >= 0; j--) for (StorageIndex p = lp_[(size_t)j]; p < lp_[(size_t)j] + lnz_[(size_t)j]; p++) x(j) -= lx_[(size_t)p] * x(li_[(size_t)p]);
    VectorType r(n);
    for (Index k = 0; k < n; k++) r(m_Pinv.indices()(k)) = x(k);                                 // P^T
    return r;
  }
  struct LNested { Index nnz; Index nonZeros() const { return nnz; } };
  struct LView { Index nnz; LNested nestedExpression() const { return LNested{nnz}; } };
  LView matrixL() const { return LView{(Index)lx_.size()}; }
  const std::vector<Scalar>& vectorD() const { return d_; }
};

}  // namespace Eigen
