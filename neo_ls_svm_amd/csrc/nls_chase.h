// Two-stage tridiagonalisation, stage 2: Hermitian band (lower bandwidth B) -> real symmetric tridiagonal by bulge chasing
// (the reference calls LAPACK's one-stage reduction at _neo_ls_svm.py:120 / :265; this is the library's own second stage).
//
// Sweep s (= column s) annihilates A[s+2 : s+B+1, s] with a reflector on rows s+1 .. s+B, applies it from both sides, and chases the
// bulge this creates down the band: stage q >= 1 of the sweep works on the off-diagonal block Bq = A[r1 : r1+B, r0 : r0+B]
// (r0 = s + 1 + (q-1) B, r1 = r0 + B) and the diagonal block Dq = A[r1 : r1+B, r1 : r1+B]:
//     Bq <- Bq H_{q-1};   H_q from the first column of Bq (annihilated below its first entry);   Bq[:, 1:] <- H_q^H Bq[:, 1:];
//     Dq <- H_q^H Dq H_q
// (only the FIRST column of a bulge is removed; the rest stays for the next sweep, so the band holds 2B sub-diagonals in flight).
// Stage q of sweep s+1 touches what stages q and q+1 of sweep s touch, so sweeps are pipelined: ONE launch of W workgroups,
// each taking the next sweep from an atomic counter (a sweep is only ever taken after its predecessor has been taken by a RUNNING
// workgroup: no deadlock whatever the residency) and waiting, stage by stage, for done[s-1] >= q + 2.
//
// Hand-off between workgroups (MI355X: private L2 per XCD, L1 never refreshed by other CUs' stores): every store of band data is an
// agent-scope (sc1, write-through) store, every storing wave drains (s_waitcnt vmcnt(0)) before the workgroup's barrier, ONE lane then
// publishes done[s] with an sc1 store; the consumer polls that word with sc1 loads from one lane, joins a barrier, and reads band
// data ONLY with sc1 loads (they bypass the L1).  One workgroup per CU (the LDS request sees to that).  No fence, no L2 write-back.
// A workgroup that waits longer than ~4 s raises ctl[1] and everybody leaves (the host reports an error): no hang.
//
// Reflectors for the back-transformation: V2[r, s] (n x n, column s = sweep s): V2[r0, s] = tau, V2[r0+1 .. r0+L-1, s] = v[1:].
#pragma once
#include "nls_sb.h"

namespace nls {
namespace chase {
using namespace trd;
using sb::one_;
using sb::zero_;

__device__ __forceinline__ double ld_sc1(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ Z ld_sc1(const Z* p) {
  const double* q = reinterpret_cast<const double*>(p);
  return {ld_sc1(q), ld_sc1(q + 1)};
}
__device__ __forceinline__ void st_sc1(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(Z* p, Z v) {
  double* q = reinterpret_cast<double*>(p);
  st_sc1(q, v.re);
  st_sc1(q + 1, v.im);
}

__host__ __device__ inline int chase_nstages(int n, int B, int s) { return s <= n - 2 ? (n - 2 - s) / B + 1 : 0; }

template <class T, int B>
struct ChaseLds {
  T Bs[B][B + 1];
  T Ds[B][B + 1];
  T v[B], vn[B], w[B], u[B], pv[B];
  T sc_t[4];        // tau (current), taun, c
  double sc_d[2];   // betan
  int sweep;
  int abort;
};

template <int TPRV, class T>
__device__ __forceinline__ T row_group_sum(T v) {  // sum over TPRV adjacent lanes (same value in all of them)
#pragma unroll
  for (int m = 1; m < TPRV; m <<= 1) v = v + shfl_xor_(v, m);
  return v;
}

// wait until done[sp] >= need (one lane polls; everybody leaves together).  Returns false on timeout / abort.
__device__ __forceinline__ bool chase_wait(const unsigned* done, int sp, unsigned need, unsigned* ctl, int* abort_slot) {
  if (threadIdx.x == 0) {
    const long long t0 = wall_clock64();
    int bad = 0;
    while (__hip_atomic_load(done + sp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
      __builtin_amdgcn_s_sleep(1);
      if (__hip_atomic_load(ctl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        bad = 1;
        break;
      }
      if (wall_clock64() - t0 > 400000000ll) {  // 4 s at 100 MHz
        __hip_atomic_store(ctl + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bad = 1;
        break;
      }
    }
    *abort_slot = bad;
  }
  __syncthreads();
  return *abort_slot == 0;
}

// every storing wave drains, the workgroup meets, one lane publishes
__device__ __forceinline__ void chase_publish(unsigned* done, int s, unsigned value) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(done + s, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Two-sided update of the Hermitian block in lds.Ds (full storage, order L) with H = I - tau v v^H (v in lds.vn, tau = taun), and store
// of its lower triangle to the band at rows / columns rb ..:  D <- H^H D H.  pv must hold taun D vn (computed by the caller's phase).
template <class T, int B>
__device__ __forceinline__ void chase_two_sided_store(ChaseLds<T, B>& lds, T taun, int L, T* AB, int ldab, long rb) {
  constexpr int TPR = 256 / B, CPT = B / TPR;
  const int r = threadIdx.x / TPR, cg = threadIdx.x % TPR;
  const int lane = threadIdx.x & 63;
  T g = zero_<T>();
  for (int i = lane; i < B; i += 64) g = g + conj_(lds.vn[i]) * lds.pv[i];
  g = wave_sum(g);  // gamma = vn^H pv, the same in every wave
  const T hg = 0.5 * (conj_(taun) * g);
  const T vr = lds.vn[r];
  const T w2r = lds.pv[r] - hg * vr;
#pragma unroll
  for (int q = 0; q < CPT; ++q) {
    const int c = cg + TPR * q;
    if (r < L && c <= r) {
      const T vc = lds.vn[c];
      const T w2c = lds.pv[c] - hg * vc;
      T x = lds.Ds[r][c] - vr * conj_(w2c) - w2r * conj_(vc);
      if (r == c) x = make_<T>(real_(x), 0.0);
      st_sc1(AB + (long)(r - c) + (rb + c) * (long)ldab, x);
    }
  }
}

template <class T, int B>
__device__ __forceinline__ void chase_load_diag(ChaseLds<T, B>& lds, const T* AB, int ldab, long rb, int L) {
  constexpr int TPR = 256 / B, CPT = B / TPR;
  const int r = threadIdx.x / TPR, cg = threadIdx.x % TPR;
#pragma unroll
  for (int q = 0; q < CPT; ++q) {
    const int c = cg + TPR * q;
    if (c <= r) {
      T x = zero_<T>();
      if (r < L) x = ld_sc1(AB + (long)(r - c) + (rb + c) * (long)ldab);
      if (r == c) x = make_<T>(real_(x), 0.0);
      lds.Ds[r][c] = x;
      if (r != c) lds.Ds[c][r] = conj_(x);
    }
  }
}

template <class T, int B>
__global__ void __launch_bounds__(256) k_chase(T* AB, int ldab, int n, T* V2, long ldv, double* d, double* e, unsigned* ctl, unsigned* done) {
  constexpr int TPR = 256 / B, CPT = B / TPR;
  extern __shared__ __attribute__((aligned(16))) unsigned char chase_smem[];
  ChaseLds<T, B>& lds = *reinterpret_cast<ChaseLds<T, B>*>(chase_smem);
  const int r = threadIdx.x / TPR, cg = threadIdx.x % TPR;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) lds.sweep = (int)atomicAdd(ctl, 1u);
    __syncthreads();
    const int s = lds.sweep;
    if (s > n - 2) return;
    const int nst = chase_nstages(n, B, s);
    const int nst_prev = s > 0 ? chase_nstages(n, B, s - 1) : 0;
    // ---------------- stage 0: column s --------------------------------------------------------
    if (s > 0 && !chase_wait(done, s - 1, (unsigned)min(2, nst_prev), ctl, &lds.abort)) return;
    long r0 = s + 1;
    int L = min(B, n - (int)r0);
    chase_load_diag<T, B>(lds, AB, ldab, r0, L);
    if (wave == 0) {
      T x = zero_<T>();
      if (lane < L) x = ld_sc1(AB + (long)(1 + lane) + (long)s * ldab);
      double xn2 = lane >= 1 ? abs2_(x) : 0.0;
      xn2 = wave_sum(xn2);
      const T alpha = lane_bcast(x, 0);
      const Larfg<T> h = larfg<T>(alpha, xn2);
      const T vi = lane == 0 ? one_<T>() : (h.identity ? zero_<T>() : x * h.scale);
      if (lane < B) {
        lds.vn[lane] = lane < L ? vi : zero_<T>();
        // the column itself: beta on the sub-diagonal, zeros below
        if (lane < L) st_sc1(AB + (long)(1 + lane) + (long)s * ldab, lane == 0 ? make_<T>(h.beta, 0.0) : zero_<T>());
        if (lane < L) V2[(r0 + lane) + (long)s * ldv] = lane == 0 ? h.tau : vi;
      }
      if (lane == 0) {
        lds.sc_t[1] = h.tau;
        e[s] = h.beta;
        d[s] = real_(ld_sc1(AB + (long)s * ldab));
      }
    }
    __syncthreads();
    T taun = lds.sc_t[1];
    {  // pv = taun D vn
      T part = zero_<T>();
#pragma unroll
      for (int q = 0; q < CPT; ++q) part = part + lds.Ds[r][cg + TPR * q] * lds.vn[cg + TPR * q];
      part = row_group_sum<TPR>(part);
      if (cg == 0) lds.pv[r] = taun * part;
    }
    __syncthreads();
    chase_two_sided_store<T, B>(lds, taun, L, AB, ldab, r0);
    if (s == n - 2 && threadIdx.x == 0) {
      // the last sweep leaves the final 1 x 1 block: d[n-1]
      const T vr = lds.vn[0];
      const T hg = 0.5 * (conj_(taun) * (conj_(vr) * lds.pv[0]));
      const T w2 = lds.pv[0] - hg * vr;
      d[n - 1] = real_(lds.Ds[0][0] - vr * conj_(w2) - w2 * conj_(vr));
    }
    chase_publish(done, s, 1u);
    // current reflector -> v, tau
    if (threadIdx.x < B) lds.v[threadIdx.x] = lds.vn[threadIdx.x];
    T tau = taun;
    // ---------------- stages q >= 1 ------------------------------------------------------------
    for (int q = 1; q < nst; ++q) {
      if (s > 0 && !chase_wait(done, s - 1, (unsigned)min(q + 2, nst_prev), ctl, &lds.abort)) return;
      __syncthreads();
      const long r1 = r0 + B;
      const int L1 = min(B, n - (int)r1);
      // loads: Bq (L1 x L) and Dq (L1 x L1, lower)
      T breg[CPT];
#pragma unroll
      for (int k = 0; k < CPT; ++k) {
        const int c = cg + TPR * k;
        T x = zero_<T>();
        if (r < L1 && c < L) x = ld_sc1(AB + (long)(B + r - c) + (r0 + c) * (long)ldab);
        breg[k] = x;
        lds.Bs[r][c] = x;
      }
      chase_load_diag<T, B>(lds, AB, ldab, r1, L1);
      __syncthreads();
      // P1: w = Bq v
      {
        T part = zero_<T>();
#pragma unroll
        for (int k = 0; k < CPT; ++k) part = part + breg[k] * lds.v[cg + TPR * k];
        part = row_group_sum<TPR>(part);
        if (cg == 0) lds.w[r] = part;
      }
      __syncthreads();
      // P2 (wave 0): first column after the right application, its reflector, c = vn^H w
      if (wave == 0) {
        T x = zero_<T>(), wi = zero_<T>();
        if (lane < B) wi = lds.w[lane];
        if (lane < L1) x = lds.Bs[lane][0] - tau * wi;  // v[0] = 1
        double xn2 = lane >= 1 ? abs2_(x) : 0.0;
        xn2 = wave_sum(xn2);
        const T alpha = lane_bcast(x, 0);
        const Larfg<T> h = larfg<T>(alpha, xn2);
        const T vi = lane == 0 ? one_<T>() : (h.identity ? zero_<T>() : x * h.scale);
        const T vnl = lane < L1 ? vi : zero_<T>();
        T cc = lane < L1 ? conj_(vnl) * wi : zero_<T>();
        cc = wave_sum(cc);
        if (lane < B) {
          lds.vn[lane] = vnl;
          if (lane < L1) V2[(r1 + lane) + (long)s * ldv] = lane == 0 ? h.tau : vi;
        }
        if (lane == 0) {
          lds.sc_t[1] = h.tau;
          lds.sc_t[2] = cc;
          lds.sc_d[0] = h.beta;
        }
      }
      __syncthreads();
      taun = lds.sc_t[1];
      // P3: u = vn^H Bq (old Bq; column dots, thread (c, rg)),  pv = taun Dq vn (row dots, thread (r, cg))
      {
        const int c = threadIdx.x / TPR, rg = threadIdx.x % TPR;
        T part = zero_<T>();
#pragma unroll
        for (int k = 0; k < CPT; ++k) part = part + conj_(lds.vn[rg + TPR * k]) * lds.Bs[rg + TPR * k][c];
        part = row_group_sum<TPR>(part);
        if (rg == 0) lds.u[c] = part;
        T p2 = zero_<T>();
#pragma unroll
        for (int k = 0; k < CPT; ++k) p2 = p2 + lds.Ds[r][cg + TPR * k] * lds.vn[cg + TPR * k];
        p2 = row_group_sum<TPR>(p2);
        if (cg == 0) lds.pv[r] = taun * p2;
      }
      __syncthreads();
      // P4: updates and stores
      {
        const T cc = lds.sc_t[2];
        const double betan = lds.sc_d[0];
        const T tw = tau * lds.w[r];
        const T tvn = conj_(taun) * lds.vn[r];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
          const int c = cg + TPR * k;
          if (r < L1 && c < L) {
            T x;
            if (c == 0) {
              x = r == 0 ? make_<T>(betan, 0.0) : zero_<T>();
            } else {
              const T vc = conj_(lds.v[c]);
              const T z = lds.u[c] - (tau * cc) * vc;
              x = breg[k] - tw * vc - tvn * z;
            }
            st_sc1(AB + (long)(B + r - c) + (r0 + c) * (long)ldab, x);
          }
        }
      }
      chase_two_sided_store<T, B>(lds, taun, L1, AB, ldab, r1);
      chase_publish(done, s, (unsigned)(q + 1));
      if (threadIdx.x < B) lds.v[threadIdx.x] = lds.vn[threadIdx.x];
      tau = taun;
      r0 = r1;
      L = L1;
    }
  }
}

}  // namespace chase
}  // namespace nls
