// One launch per column of the Householder tridiagonalisation (the serial section of P4: eigh(A / c), _neo_ls_svm.py:120; D2: :265).
//
// nls_trd.h runs a column as TWO dependent launches: k_trd_hemv2 (matrix-vector product on the lower-triangle tiles + the dot blocks) and
// k_trd_finish2 (sum of the strip partials, larfg scalars, v, w', the base of the next column).  At n = 4097 the second launch is 7.5 us x 4096
// columns = 31 ms of which the kernel boundary itself - the GPU's dependent-dispatch latency - is more than half (VERDICT r04 weak #3, next #3d);
// at n = 1025 it is 7.7 of the 13 ms the reduction takes.  Here the finish blocks ride in the SAME grid, behind the producers (tile and dot blocks):
//   * a producer writes its partials with agent-scope (sc1, write-through) stores, drains them (s_waitcnt vmcnt(0)), meets its workgroup at a
//     barrier, and ONE lane publishes flags[block] = epoch (epoch = column + 1: monotone, nothing is ever reset inside a reduction);
//   * a finish block first issues every load that does not depend on this column's producers, then polls - one flag per thread - the flags of
//     exactly the blocks it needs: the dot blocks (norm / W^H x / V^H x partials, x itself) and the tiles of ITS row strip and column strip
//     (<= 2 x 65 + 65 flags at n = 4097), and reads the partials with sc1 loads (they bypass the L1, which other CUs' stores never refresh).
//     This is the hand-off nls_chase.h uses between the workgroups of the bulge chase; no fence, no L2 write-back, no grid-wide barrier.
//   * no deadlock: producers never wait for anything, and a finish block is dispatched only after every producer of its queue has been
//     (workgroups leave a queue in index order; the finish blocks have the highest indices).  A poll that lasts longer than ~0.5 s raises the
//     error word and the block carries on with what it has (the host then fails the call): no hang.
// Everything the finish blocks produce (v, w', the base of column j + 1, the w'^H v partials, e, tau) is consumed by the NEXT launch: plain stores.
// Arithmetic and summation orders are those of k_trd_hemv2 / k_trd_finish2: results are bit-identical to the two-launch form (tests).
#pragma once
#include "nls_chase.h"

namespace nls {
namespace trd {

struct Fuse {
  unsigned* flags;  // one word per producer block of a column
  unsigned* err;    // raised by a finish block whose poll timed out
  unsigned epoch;   // column + 1
};

template <class T>
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const T* p, long count) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(p), 0, (int)(count * (long)sizeof(T)), 0x00020000);
}
template <class T>
__device__ __forceinline__ T ld1(__amdgpu_buffer_rsrc_t rs, long idx) {  // element idx through the descriptor, agent scope
  return chase::chase_ld(rs, (unsigned)(idx * (long)sizeof(T)), T());
}
template <class T>
__device__ __forceinline__ void st1(__amdgpu_buffer_rsrc_t rs, long idx, T v) {
  chase::chase_st(rs, (unsigned)(idx * (long)sizeof(T)), v);
}
__device__ __forceinline__ double ld1d(const double* p) { return chase::ld_sc1(p); }

// ---- producers: the body of k_trd_hemv2 with agent-scope stores and a published flag ---------------------------------------------------------
template <class T>
__device__ __forceinline__ void trd_col1_producer(const Args<T>& a, int S0, int ntiles, int bid, const Fuse& f) {
  __shared__ T sh[4][TS];
  __shared__ T slot;
  const int n = a.n, j = a.j, i = a.j - a.j0;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int NSC = (n + TS - 1) / TS, NSR = (n + RT - 1) / RT;
  const __amdgpu_buffer_rsrc_t rs_ylow = rsrc_of(a.ylow, (long)NSC * n), rs_yup = rsrc_of(a.yup, (long)NSR * n);
  PrevScalars<T> ps;
  ps.mu = ps.alpha2 = make_<T>(0.0, 0.0);
  if (i > 0) ps = prev_scalars(a, &slot);
  auto xval = [&](long q) -> T {
    if (q <= j || q >= n) return make_<T>(0.0, 0.0);
    return i > 0 ? a.bvec[q] + ps.mu * a.A[q + (long)(j - 1) * a.lda] : a.A[q + (long)j * a.lda];
  };
  if (bid < ntiles) {
    int t = (a.boustrophedon && (a.j & 1)) ? ntiles - 1 - bid : bid, Rr = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
    while ((Rr + 1) * (Rr + 2) / 2 <= t) ++Rr;
    while (Rr * (Rr + 1) / 2 > t) --Rr;
    const int R = S0 + Rr, C = S0 + (t - Rr * (Rr + 1) / 2);
    const long r = (long)R * RT + lane;
    T av[2 * GW];
#pragma unroll
    for (int cc = 0; cc < 2 * GW; ++cc) {
      const long c = (long)C * TS + 2 * GW * w + cc;
      av[cc] = make_<T>(0.0, 0.0);
      if (r < n && c < n && r >= c) av[cc] = a.A[r + c * a.lda];
    }
    const T xr = xval(r);
    const T xcol = xval((long)C * TS + 2 * GW * w + (lane & 15));
    T low = make_<T>(0.0, 0.0);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      T up[GW];
#pragma unroll
      for (int cc = 0; cc < GW; ++cc) {
        const long c = (long)C * TS + 2 * GW * w + GW * g + cc;
        T v = av[GW * g + cc];
        if (r == c) v = make_<T>(real_(v), 0.0);
        const T xc = lane_bcast(xcol, GW * g + cc);
        low = low + v * xc;
        up[cc] = sel_(r > c, conj_(v) * xr, make_<T>(0.0, 0.0));
      }
      butterfly8(up, lane);
      if ((lane & 7) == 0) {
        const long c = (long)C * TS + 2 * GW * w + GW * g + butterfly_col(lane);
        if (c < n) st1(rs_yup, (long)R * n + c, up[0]);
      }
    }
    sh[w][lane] = low;
    __syncthreads();
    if (w == 0 && r < n) st1(rs_ylow, (long)C * n + r, ((sh[0][lane] + sh[1][lane]) + sh[2][lane]) + sh[3][lane]);
  } else {
    const int b = bid - ntiles;
    const long r = (long)j + (long)b * RD + lane;
    const bool inrange = r < n;
    const __amdgpu_buffer_rsrc_t rs_x = rsrc_of(a.xvec, (long)n), rs_W = rsrc_of(a.W, (long)n * NB), rs_z = rsrc_of(a.zpart, (long)a.ndot * 2 * NB);
    T xfull = make_<T>(0.0, 0.0), wt = make_<T>(0.0, 0.0), vprev = make_<T>(0.0, 0.0);
    if (inrange) {
      if (i > 0) {
        vprev = a.A[r + (long)(j - 1) * a.lda];
        wt = a.wtmp_prev[r];
        xfull = a.bvec[r] + ps.mu * vprev;
      } else {
        xfull = a.A[r + (long)j * a.lda];
      }
    }
    const bool live = inrange && r > j;
    const T xr = live ? xfull : make_<T>(0.0, 0.0);
    const T wlast = wt + ps.alpha2 * vprev;  // final W[r][i - 1]
    T mw[GW], mv[GW];
#pragma unroll
    for (int k = 0; k < GW; ++k) {
      const int p = w + 4 * k;
      mw[k] = mv[k] = make_<T>(0.0, 0.0);
      if (live && p < i) {
        mw[k] = p == i - 1 ? wlast : a.W[r + (long)p * n];
        mv[k] = a.A[r + (long)(a.j0 + p) * a.lda];
      }
    }
    if (w == 0 && inrange) {
      if (r == j) {
        const double dj = real_(xfull);
        a.d[j] = dj;
        a.A[r + (long)j * a.lda] = make_<T>(dj, 0.0);
      } else {
        st1(rs_x, r, xfull);
      }
      if (i > 0) st1(rs_W, r + (long)(i - 1) * n, wlast);
    }
    if (w == 0) {
      const double nrm = wave_sum(inrange && r >= j + 2 ? abs2_(xfull) : 0.0);
      if (lane == 0) chase::st_sc1(a.pnorm + b, nrm);
    }
#pragma unroll
    for (int k = 0; k < GW; ++k) {
      mw[k] = conj_(mw[k]) * xr;
      mv[k] = conj_(mv[k]) * xr;
    }
    butterfly8(mw, lane);
    butterfly8(mv, lane);
    if ((lane & 7) == 0) {
      const int p = w + 4 * butterfly_col(lane);
      st1(rs_z, (long)b * 2 * NB + p, sel_(p < i, mw[0], make_<T>(0.0, 0.0)));
      st1(rs_z, (long)b * 2 * NB + NB + p, sel_(p < i, mv[0], make_<T>(0.0, 0.0)));
    }
  }
  // every storing wave drains, the workgroup meets, one lane publishes
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(f.flags + bid, f.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- finish blocks: the body of k_trd_finish2 behind a poll of the producers' flags ---------------------------------------------------------
template <class T>
__device__ __forceinline__ void trd_col1_finish(const Args<T>& a, int S0, int NS, int ntiles, int bid, const Fuse& f) {
  __shared__ T zsh[4][2 * NB], zw[NB], zv[NB], wj1[NB], vj1[NB], red[ROWT];
  __shared__ double dslot;
  constexpr int PPT = NB / TPR;
  const int n = a.n, j = a.j, i = a.j - a.j0;
  const int q = threadIdx.x % TPR, rl = threadIdx.x / TPR;
  const long r = (long)bid * ROWT + rl;
  const bool live = r < n && r >= j + 1;
  if ((long)bid * ROWT + ROWT - 1 < j + 1 && bid != 0) {  // every row of the block lies above the trailing matrix: nothing to wait for, nothing to finish
    if (threadIdx.x == 0) a.spart[bid] = make_<T>(0.0, 0.0);
    return;
  }
  const int NSC = (n + TS - 1) / TS;
  const __amdgpu_buffer_rsrc_t rs_ylow = rsrc_of(a.ylow, (long)NSC * n), rs_yup = rsrc_of(a.yup, (long)NS * n), rs_x = rsrc_of(a.xvec, (long)n),
                               rs_W = rsrc_of(a.W, (long)n * NB), rs_z = rsrc_of(a.zpart, (long)a.ndot * 2 * NB);
  // ---- loads that do not depend on this column's producers (in flight during the poll)
  T zrow = make_<T>(0.0, 0.0);  // row j + 1 of W (slots < NB) and of V; W's column i - 1 is finished by this column's dot blocks: after the poll
  const bool zrow_late = threadIdx.x < NB && threadIdx.x == i - 1;
  if (threadIdx.x < 2 * NB && threadIdx.x % NB < i && !zrow_late) {
    const int p = threadIdx.x % NB;
    zrow = threadIdx.x >= NB ? a.A[(long)(j + 1) + (long)(a.j0 + p) * a.lda] : a.W[(long)(j + 1) + (long)p * n];
  }
  T y = make_<T>(0.0, 0.0), xr = make_<T>(0.0, 0.0), t0 = make_<T>(0.0, 0.0), vv[PPT], ww[PPT];
#pragma unroll
  for (int k = 0; k < PPT; ++k) vv[k] = ww[k] = make_<T>(0.0, 0.0);
  if (live) {
    t0 = a.A[r + (long)(j + 1) * a.lda];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int p = q + TPR * k;
      if (p < i) {
        vv[k] = a.A[r + (long)(a.j0 + p) * a.lda];
        if (p < i - 1) ww[k] = a.W[r + (long)p * n];
      }
    }
  }
  // ---- poll: thread t watches one producer.  Needed: every dot block; the tiles (Rb, C <= Rb) and (R >= Rb, Rb) of this block's strip Rb.
  {
    const int Rb = (int)(((long)bid * ROWT) / RT);
    const int K = NS - S0;            // active strips
    const int rb = Rb - S0;           // strip index among the active ones (< 0: rows above the trailing matrix - no tiles needed)
    const int nrow = rb >= 0 ? rb + 1 : 0, ncol = rb >= 0 ? K - rb - 1 : 0;  // tiles (rb, 0 .. rb) and (rb + 1 .. K - 1, rb)
    const int t = threadIdx.x;
    int watch = -1;
    if (t < a.ndot) {
      watch = ntiles + t;
    } else if (ntiles > 0 && t - a.ndot < nrow + ncol) {
      const int u = t - a.ndot;
      const int Rr = u < nrow ? rb : rb + 1 + (u - nrow), Cc = u < nrow ? u : rb;
      const int tt = Rr * (Rr + 1) / 2 + Cc;
      watch = (a.boustrophedon && (a.j & 1)) ? ntiles - 1 - tt : tt;
    }
    if (watch >= 0) {
      const long long t0c = wall_clock64();
      while (__hip_atomic_load(f.flags + watch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != f.epoch) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0c > 50000000ll) {  // 0.5 s at 100 MHz
          __hip_atomic_store(f.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __syncthreads();
  }
  // ---- loads of what the producers wrote (agent scope)
  double pn = 0.0;
  if (threadIdx.x < 64)
    for (int b = threadIdx.x; b < a.ndot; b += 64) pn += ld1d(a.pnorm + b);
  const T alpha = ld1<T>(rs_x, j + 1);
  T zp = make_<T>(0.0, 0.0);
  {
    const int slot = threadIdx.x % (2 * NB), part = threadIdx.x / (2 * NB);
    if (slot % NB < i)
      for (int b = part; b < a.ndot; b += 4) zp = zp + ld1<T>(rs_z, (long)b * 2 * NB + slot);
  }
  if (zrow_late) zrow = ld1<T>(rs_W, (long)(j + 1) + (long)(i - 1) * n);
  if (live) {
    T y1 = make_<T>(0.0, 0.0), y2 = y1, y3 = y1;
    const int Clast = (int)(r / TS);
    for (int C = S0 + q; C <= Clast; C += 4 * TPR) {
      y = y + ld1<T>(rs_ylow, (long)C * n + r);
      if (C + TPR <= Clast) y1 = y1 + ld1<T>(rs_ylow, (long)(C + TPR) * n + r);
      if (C + 2 * TPR <= Clast) y2 = y2 + ld1<T>(rs_ylow, (long)(C + 2 * TPR) * n + r);
      if (C + 3 * TPR <= Clast) y3 = y3 + ld1<T>(rs_ylow, (long)(C + 3 * TPR) * n + r);
    }
    for (int R = (int)(r / RT) + q; R < NS; R += 4 * TPR) {
      y = y + ld1<T>(rs_yup, (long)R * n + r);
      if (R + TPR < NS) y1 = y1 + ld1<T>(rs_yup, (long)(R + TPR) * n + r);
      if (R + 2 * TPR < NS) y2 = y2 + ld1<T>(rs_yup, (long)(R + 2 * TPR) * n + r);
      if (R + 3 * TPR < NS) y3 = y3 + ld1<T>(rs_yup, (long)(R + 3 * TPR) * n + r);
    }
    y = (y + y1) + (y2 + y3);
    xr = ld1<T>(rs_x, r);
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
      const int p = q + TPR * k;
      if (p < i && p == i - 1) ww[k] = ld1<T>(rs_W, r + (long)p * n);
    }
  }
  // ---- from here on: k_trd_finish2, unchanged
  if (threadIdx.x < 64) {
    pn = wave_sum(pn);
    if (threadIdx.x == 0) dslot = pn;
  }
  zsh[threadIdx.x / (2 * NB)][threadIdx.x % (2 * NB)] = zp;
  __syncthreads();
  const Larfg<T> h = larfg<T>(alpha, dslot);
  const T beta = make_<T>(h.beta, 0.0);
  if (threadIdx.x < 2 * NB) {
    const int p = threadIdx.x % NB;
    T z = make_<T>(0.0, 0.0);
    if (p < i && !h.identity)
      z = ((((zsh[0][threadIdx.x] + zsh[1][threadIdx.x]) + zsh[2][threadIdx.x]) + zsh[3][threadIdx.x]) - beta * conj_(zrow)) * h.scale;
    if (threadIdx.x >= NB) {
      zv[p] = z;
      vj1[p] = conj_(zrow);
    } else {
      zw[p] = z;
      wj1[p] = conj_(zrow);
    }
  }
  __syncthreads();
  T corr = make_<T>(0.0, 0.0), upd = make_<T>(0.0, 0.0);
#pragma unroll
  for (int k = 0; k < PPT; ++k) {
    const int p = q + TPR * k;
    if (live && p < i) {
      corr = corr + vv[k] * zw[p] + ww[k] * zv[p];
      upd = upd + vv[k] * wj1[p] + ww[k] * vj1[p];
    }
  }
  y = row_sum(y);
  corr = row_sum(corr);
  upd = row_sum(upd);
  T sv = make_<T>(0.0, 0.0);
  if (live && q == 0) {
    T v, wp = make_<T>(0.0, 0.0);
    T t0d = t0;
    if (r == j + 1) t0d = make_<T>(real_(t0), 0.0);
    if (h.identity) {
      v = make_<T>(r == j + 1 ? 1.0 : 0.0, 0.0);
    } else {
      v = r == j + 1 ? make_<T>(1.0, 0.0) : xr * h.scale;
      wp = h.tau * ((y - beta * t0d) * h.scale - corr);
    }
    a.wtmp[r] = wp;
    a.A[r + (long)j * a.lda] = v;
    if (a.make_base) a.bvec[r] = t0d - upd - wp;
    sv = conj_(wp) * v;
  }
  if (r == 0 && q == 0) {
    a.e[j] = h.beta;
    a.tau[j] = h.tau;
  }
  if (q == 0) red[rl] = sv;
  __syncthreads();
  if (threadIdx.x < 64) {
    const T t = wave_sum(sel_(threadIdx.x < ROWT, red[threadIdx.x % ROWT], make_<T>(0.0, 0.0)));
    if (threadIdx.x == 0) a.spart[bid] = t;
  }
}

// grid = ntiles + ndot producers, then nrowblocks finish blocks; 256 threads.
template <class T>
__global__ void __launch_bounds__(256, 2) k_trd_col1(Args<T> a, int S0, int ntiles, int NS, Fuse f) {
  const int nprod = ntiles + a.ndot;
  if ((int)blockIdx.x < nprod)
    trd_col1_producer<T>(a, S0, ntiles, (int)blockIdx.x, f);
  else
    trd_col1_finish<T>(a, S0, NS, ntiles, (int)blockIdx.x - nprod, f);
}

}  // namespace trd
}  // namespace nls
