// ORACLE — TEST INFRASTRUCTURE ONLY (see functors.hpp header).  PARITY UNPINNED by the reference.
//
// sparse_chol.hpp — exact sparse Cholesky on a symmetric positive-definite matrix of dense 6x6 blocks,
// standing in for Ceres' `SPARSE_NORMAL_CHOLESKY` (reference src/PoseGraphSLAM.cpp:1270).  Ceres delegates
// that factorisation to SuiteSparse/CXSparse/Eigen (not in /root/reference, unpinned); any exact Cholesky
// yields the same solution up to rounding, so this restates the textbook algorithm:
//   * fill-reducing ordering: approximate minimum degree on the block graph (quotient graph with element
//     absorption and Amestoy-Davis-Duff approximate external degrees; no supervariables)
//   * elimination tree + up-looking numeric factorisation (the CSparse `cs_chol` scheme) on 6x6 blocks
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <set>
#include <vector>

namespace orc {

// ---- small dense 6x6 helpers (row-major) ----
inline void b6_zero(double* A) { std::memset(A, 0, 36 * sizeof(double)); }
// C -= A * B^T
inline void b6_submul_abt(double* C, const double* A, const double* B) {
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) {
            double s = 0;
            for (int k = 0; k < 6; ++k) s += A[i * 6 + k] * B[j * 6 + k];
            C[i * 6 + j] -= s;
        }
}
// in-place lower Cholesky of a 6x6 SPD block; returns false if not PD
inline bool b6_chol(double* A) {
    for (int j = 0; j < 6; ++j) {
        double d = A[j * 6 + j];
        for (int k = 0; k < j; ++k) d -= A[j * 6 + k] * A[j * 6 + k];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d);
        A[j * 6 + j] = d;
        for (int i = j + 1; i < 6; ++i) {
            double s = A[i * 6 + j];
            for (int k = 0; k < j; ++k) s -= A[i * 6 + k] * A[j * 6 + k];
            A[i * 6 + j] = s / d;
        }
        for (int i = 0; i < j; ++i) A[i * 6 + j] = 0.0;
    }
    return true;
}
// X <- X * L^-T   (L lower 6x6): solves Y L^T = X row by row
inline void b6_right_solve_lt(double* X, const double* L) {
    for (int r = 0; r < 6; ++r) {
        double* x = X + r * 6;
        for (int j = 0; j < 6; ++j) {
            double s = x[j];
            for (int k = 0; k < j; ++k) s -= x[k] * L[j * 6 + k];
            x[j] = s / L[j * 6 + j];
        }
    }
}

// ---- approximate minimum degree ordering on an undirected graph given as adjacency lists ----
inline std::vector<int> amd_order(const std::vector<std::vector<int>>& adj_in) {
    const int n = (int)adj_in.size();
    std::vector<std::vector<int>> A(adj_in);       // variable-variable adjacency (pruned as we go)
    std::vector<std::vector<int>> E(n);            // variable -> adjacent elements
    std::vector<std::vector<int>> L(n);            // element -> variables (element id = pivot variable id)
    std::vector<char> is_elim(n, 0), elem_alive(n, 0);
    std::vector<int> deg(n), w(n, -1), mark(n, -1), perm;
    std::vector<int> wstamp(n, -1);
    perm.reserve(n);
    std::set<std::pair<int, int>> pq;
    for (int i = 0; i < n; ++i) {
        std::sort(A[i].begin(), A[i].end());
        A[i].erase(std::unique(A[i].begin(), A[i].end()), A[i].end());
        A[i].erase(std::remove(A[i].begin(), A[i].end(), i), A[i].end());
        deg[i] = (int)A[i].size();
        pq.insert({deg[i], i});
    }
    std::vector<int> Lp;
    for (int k = 0; k < n; ++k) {
        const int p = pq.begin()->second;
        pq.erase(pq.begin());
        is_elim[p] = 1;
        perm.push_back(p);
        // Lp = (A_p U union L_e for e in E_p) \ {p}
        Lp.clear();
        mark[p] = k;
        for (int v : A[p]) if (!is_elim[v] && mark[v] != k) { mark[v] = k; Lp.push_back(v); }
        for (int e : E[p]) {
            if (!elem_alive[e]) continue;
            for (int v : L[e]) if (!is_elim[v] && mark[v] != k) { mark[v] = k; Lp.push_back(v); }
            elem_alive[e] = 0;  // absorbed into p
            std::vector<int>().swap(L[e]);
        }
        std::vector<int>().swap(A[p]);
        std::vector<int>().swap(E[p]);
        L[p] = Lp;
        elem_alive[p] = 1;
        const int lp = (int)Lp.size();
        // w[e] = |L_e \ Lp| for elements touching Lp
        for (int i : Lp) {
            auto& Ei = E[i];
            size_t o = 0;
            for (size_t q = 0; q < Ei.size(); ++q) {
                const int e = Ei[q];
                if (!elem_alive[e]) continue;  // drop absorbed
                Ei[o++] = e;
                if (wstamp[e] != k) { wstamp[e] = k; w[e] = (int)L[e].size(); }
                --w[e];
            }
            Ei.resize(o);
        }
        for (int i : Lp) {
            // prune A_i: drop eliminated vars and members of Lp (now covered by element p)
            auto& Ai = A[i];
            size_t o = 0;
            for (size_t q = 0; q < Ai.size(); ++q) {
                const int v = Ai[q];
                if (is_elim[v] || mark[v] == k) continue;
                Ai[o++] = v;
            }
            Ai.resize(o);
            // aggressive absorption + approximate degree
            auto& Ei = E[i];
            long long d = (long long)Ai.size() + (lp - 1);
            o = 0;
            for (size_t q = 0; q < Ei.size(); ++q) {
                const int e = Ei[q];
                if (!elem_alive[e]) continue;
                if (w[e] == 0) {  // L_e subset of Lp: absorb e into p
                    continue;     // (element freed lazily below)
                }
                Ei[o++] = e;
                d += w[e];
            }
            Ei.resize(o);
            Ei.push_back(p);
            const long long bound1 = (long long)(n - k - 1);
            const long long bound2 = (long long)deg[i] + (lp - 1);
            long long nd = std::min(d, std::min(bound1, bound2));
            if (nd < 0) nd = 0;
            pq.erase({deg[i], i});
            deg[i] = (int)nd;
            pq.insert({deg[i], i});
        }
        // free aggressively absorbed elements
        for (int i : Lp) (void)i;
    }
    return perm;
}

// Block-sparse SPD matrix in coordinate form: diagonal blocks + strictly-lower/upper pairs (i != j).
struct BlockSPD {
    int n = 0;                         // block dimension
    std::vector<double> diag;          // n*36
    std::vector<int> oi, oj;           // off-diagonal block (oi, oj), value = block at row oi, col oj; its transpose is implied
    std::vector<double> oval;          // 36 per off-diagonal block.  Duplicates (same pair) are summed.
};

struct BlockCholesky {
    int n = 0;
    std::vector<int> perm, iperm, parent;
    std::vector<int64_t> colptr;       // n+1, block CSC of L (diagonal first in each column)
    std::vector<int> rowind;
    std::vector<double> val;           // 36 per block
    int64_t nnz_blocks = 0;
    double flops = 0;
    bool symbolic_only = false;        // analyze_and_factor stops after the symbolic phase: ordering, elimination tree, column counts (nnz_blocks) and the flop count of the numeric phase

    // upper-triangular permuted A in block CSC: column k holds rows i<=k
    std::vector<int64_t> acolptr; std::vector<int> arow; std::vector<double> aval;

    bool analyze_and_factor(const BlockSPD& A, const std::vector<int>* fixed_perm = nullptr) {
        n = A.n;
        // ---- ordering
        if (fixed_perm) perm = *fixed_perm;
        else {
            std::vector<std::vector<int>> adj(n);
            for (size_t e = 0; e < A.oi.size(); ++e) { adj[A.oi[e]].push_back(A.oj[e]); adj[A.oj[e]].push_back(A.oi[e]); }
            perm = amd_order(adj);
        }
        iperm.assign(n, 0);
        for (int k = 0; k < n; ++k) iperm[perm[k]] = k;
        // ---- permuted upper-triangular block CSC (sum duplicates)
        struct Ent { int r, c; int64_t src; bool transposed; };
        std::vector<Ent> ents;
        ents.reserve(A.oi.size() + n);
        for (int i = 0; i < n; ++i) ents.push_back({iperm[i], iperm[i], (int64_t)i, false});
        for (size_t e = 0; e < A.oi.size(); ++e) {
            int r = iperm[A.oi[e]], c = iperm[A.oj[e]];
            bool tr = false;
            if (r > c) { std::swap(r, c); tr = true; }
            ents.push_back({r, c, (int64_t)e + n, tr});   // src >= n marks off-diagonal
        }
        std::sort(ents.begin(), ents.end(), [](const Ent& a, const Ent& b) { return a.c != b.c ? a.c < b.c : (a.r != b.r ? a.r < b.r : a.src < b.src); });
        acolptr.assign(n + 1, 0); arow.clear(); aval.clear();
        {
            int lastc = -1, lastr = -1;
            for (const Ent& en : ents) {
                if (en.c != lastc || en.r != lastr) {
                    arow.push_back(en.r);
                    aval.resize(aval.size() + 36, 0.0);
                    acolptr[en.c + 1]++;
                    lastc = en.c; lastr = en.r;
                }
                double* dst = &aval[aval.size() - 36];
                const double* src = en.src < n ? &A.diag[en.src * 36] : &A.oval[(en.src - n) * 36];
                if (!en.transposed) for (int q = 0; q < 36; ++q) dst[q] += src[q];
                else for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) dst[a * 6 + b] += src[b * 6 + a];
            }
            for (int c = 0; c < n; ++c) acolptr[c + 1] += acolptr[c];
        }
        // ---- elimination tree (cs_etree on upper-triangular CSC)
        parent.assign(n, -1);
        {
            std::vector<int> ancestor(n, -1);
            for (int k = 0; k < n; ++k) {
                for (int64_t p = acolptr[k]; p < acolptr[k + 1]; ++p) {
                    int i = arow[p];
                    while (i != -1 && i < k) {
                        const int inext = ancestor[i];
                        ancestor[i] = k;
                        if (inext == -1) parent[i] = k;
                        i = inext;
                    }
                }
            }
        }
        // ---- symbolic: column counts via ereach per row
        std::vector<int> flag(n, -1), stack(n), colcount(n, 1);
        auto ereach = [&](int k, int* s) -> int {  // returns top; pattern in s[top..n-1], topological order
            int top = n;
            flag[k] = k;
            for (int64_t p = acolptr[k]; p < acolptr[k + 1]; ++p) {
                int i = arow[p];
                if (i > k) continue;
                int len = 0;
                for (; flag[i] != k; i = parent[i]) { stack[len++] = i; flag[i] = k; }
                while (len > 0) s[--top] = stack[--len];
            }
            return top;
        };
        std::vector<int> s(n);
        for (int k = 0; k < n; ++k) {
            const int top = ereach(k, s.data());
            for (int q = top; q < n; ++q) colcount[s[q]]++;
        }
        colptr.assign(n + 1, 0);
        for (int c = 0; c < n; ++c) colptr[c + 1] = colptr[c] + colcount[c];
        nnz_blocks = colptr[n];
        if (symbolic_only) {      // the flops the numeric loop below would spend (its own count: 2 * 216 per block update), without touching a value
            std::fill(flag.begin(), flag.end(), -1);
            std::vector<int64_t> cnt(n, 1);      // blocks of column i so far (the diagonal block is there from the start)
            flops = 0;
            for (int k = 0; k < n; ++k) {
                const int top = ereach(k, s.data());
                for (int q = top; q < n; ++q) { const int i = s[q]; flops += 2.0 * 216.0 * (double)cnt[i]; ++cnt[i]; }
            }
            return true;
        }
        rowind.assign(nnz_blocks, 0);
        val.assign((size_t)nnz_blocks * 36, 0.0);
        // ---- numeric, up-looking
        std::fill(flag.begin(), flag.end(), -1);
        std::vector<int64_t> next(colptr.begin(), colptr.end() - 1);  // next free slot per column
        std::vector<double> x((size_t)n * 36, 0.0);
        double D[36];
        flops = 0;
        for (int k = 0; k < n; ++k) {
            const int top = ereach(k, s.data());
            // scatter A(0:k,k)
            b6_zero(D);
            for (int64_t p = acolptr[k]; p < acolptr[k + 1]; ++p) {
                const int i = arow[p];
                if (i < k) {
                    // x[i] must hold A_ki = A_ik^T
                    double* xi = &x[(size_t)i * 36];
                    const double* a = &aval[p * 36];
                    for (int r = 0; r < 6; ++r) for (int c = 0; c < 6; ++c) xi[r * 6 + c] = a[c * 6 + r];
                } else if (i == k) {
                    std::memcpy(D, &aval[p * 36], 36 * sizeof(double));
                }
            }
            for (int q = top; q < n; ++q) {
                const int i = s[q];
                double* xi = &x[(size_t)i * 36];
                // L_ki = x_i * L_ii^-T
                const double* Lii = &val[colptr[i] * 36];
                b6_right_solve_lt(xi, Lii);
                // propagate to later rows of column i
                for (int64_t p = colptr[i] + 1; p < next[i]; ++p) {
                    const int r = rowind[p];
                    b6_submul_abt(&x[(size_t)r * 36], xi, &val[p * 36]);   // x_r -= L_ki * L_ri^T
                }
                flops += 2.0 * 216.0 * (double)(next[i] - colptr[i]);
                b6_submul_abt(D, xi, xi);                                   // D -= L_ki L_ki^T
                // store L_ki in column i
                const int64_t slot = next[i]++;
                rowind[slot] = k;
                std::memcpy(&val[slot * 36], xi, 36 * sizeof(double));
                b6_zero(xi);
            }
            // symmetrise D's lower part and factor
            if (!b6_chol(D)) return false;
            const int64_t slot = next[k]++;
            rowind[slot] = k;
            std::memcpy(&val[slot * 36], D, 36 * sizeof(double));
        }
        return true;
    }

    // Solves A y = b in place (b of size 6n, original ordering).
    void solve(double* b) const {
        std::vector<double> z((size_t)n * 6);
        for (int k = 0; k < n; ++k) std::memcpy(&z[(size_t)k * 6], &b[(size_t)perm[k] * 6], 6 * sizeof(double));
        // forward: L z' = z (column oriented)
        for (int j = 0; j < n; ++j) {
            double* zj = &z[(size_t)j * 6];
            const double* Ljj = &val[colptr[j] * 36];
            for (int r = 0; r < 6; ++r) {
                double sacc = zj[r];
                for (int c = 0; c < r; ++c) sacc -= Ljj[r * 6 + c] * zj[c];
                zj[r] = sacc / Ljj[r * 6 + r];
            }
            for (int64_t p = colptr[j] + 1; p < colptr[j + 1]; ++p) {
                double* zr = &z[(size_t)rowind[p] * 6];
                const double* Lrj = &val[p * 36];
                for (int r = 0; r < 6; ++r) {
                    double sacc = 0;
                    for (int c = 0; c < 6; ++c) sacc += Lrj[r * 6 + c] * zj[c];
                    zr[r] -= sacc;
                }
            }
        }
        // backward: L^T y = z
        for (int j = n - 1; j >= 0; --j) {
            double* zj = &z[(size_t)j * 6];
            for (int64_t p = colptr[j] + 1; p < colptr[j + 1]; ++p) {
                const double* zr = &z[(size_t)rowind[p] * 6];
                const double* Lrj = &val[p * 36];
                for (int c = 0; c < 6; ++c) {
                    double sacc = 0;
                    for (int r = 0; r < 6; ++r) sacc += Lrj[r * 6 + c] * zr[r];
                    zj[c] -= sacc;
                }
            }
            const double* Ljj = &val[colptr[j] * 36];
            for (int c = 5; c >= 0; --c) {
                double sacc = zj[c];
                for (int r = c + 1; r < 6; ++r) sacc -= Ljj[r * 6 + c] * zj[r];
                zj[c] = sacc / Ljj[c * 6 + c];
            }
        }
        for (int k = 0; k < n; ++k) std::memcpy(&b[(size_t)perm[k] * 6], &z[(size_t)k * 6], 6 * sizeof(double));
    }
};

}  // namespace orc
