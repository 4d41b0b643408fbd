// spatial_box_box.hpp -- link-link box contacts of the tree kernel: ODE dBoxBox restated for one lane per shape pair.
// Part of the gfx950 tree kernel; overview in spatial_kernel.hpp, design in DESIGN.md section 4.2.
//
// Attribution: the two routines below (sp_clip_rect_quad, sp_box_box) are a per-lane rewrite of a PUBLISHED algorithm -- the
// box-box collider of the Open Dynamics Engine, ode/src/box.cpp: `intersectRectQuad` and `dBoxBox` -- the third-party routine DART's
// ODE collision detector calls for two boxes (the detector the reference selects, gym/envs/dart/walker3d.py:26 + dart_env.py).  They
// follow ODE's procedure step for step (axis order, the 1.05 face preference, the clipping order, contact positions) because the
// contact set has to be ODE's for parity; they are not an independent derivation.
//   Open Dynamics Engine, Copyright (C) 2001-2003 Russell L. Smith.  All rights reserved.  ODE is dual-licensed under the GNU LGPL
//   (2.1 or later) and a BSD-style license; this file uses it under the BSD-style license:
//   Redistribution and use in source and binary forms, with or without modification, are permitted provided that the following
//   conditions are met: (1) redistributions of source code must retain the above copyright notice, this list of conditions and the
//   following disclaimer; (2) redistributions in binary form must reproduce the above copyright notice, this list of conditions and
//   the following disclaimer in the documentation and/or other materials provided with the distribution; (3) neither the names of
//   ODE's copyright owner nor the names of its contributors may be used to endorse or promote products derived from this software
//   without specific prior written permission.  THIS SOFTWARE IS PROVIDED BY THE COPYRIGHT HOLDERS AND CONTRIBUTORS "AS IS" AND ANY
//   EXPRESS OR IMPLIED WARRANTIES, INCLUDING, BUT NOT LIMITED TO, THE IMPLIED WARRANTIES OF MERCHANTABILITY AND FITNESS FOR A
//   PARTICULAR PURPOSE ARE DISCLAIMED.  IN NO EVENT SHALL THE COPYRIGHT OWNER OR CONTRIBUTORS BE LIABLE FOR ANY DIRECT, INDIRECT,
//   INCIDENTAL, SPECIAL, EXEMPLARY, OR CONSEQUENTIAL DAMAGES ARISING IN ANY WAY OUT OF THE USE OF THIS SOFTWARE.
#pragma once
#include "spatial_model.hpp"

namespace dartk {

// ------------------------------------------------------------------ box-box contacts between two links
// ODE's dBoxBox (the routine behind DART's ODE detector for two boxes): separating-axis test over the 15 axes with the
// 1.05 preference for face axes; edge-edge -> one point midway between the closest points of the two edges; face case
// -> the incident face of the other box is clipped against the reference face's rectangle and the clipped vertices
// below the reference face are the contacts.  Returns the number of points (<= 8) written as (x, y, z, depth) to `out`
// (40 Reals of LDS workspace); `normal` points from the first box to the second.
template <class Real>
__device__ __forceinline__ int sp_clip_rect_quad(const Real* h, Real* p, Real* ret, Real* buffer) {
  int nq = 4, nr = 0;
  Real* q = p;
  Real* r = ret;
  for (int dir = 0; dir <= 1; dir++) {
    for (int sign = -1; sign <= 1; sign += 2) {
      Real* pq = q;
      Real* pr = r;
      nr = 0;
      bool full = false;
      for (int i = nq; i > 0 && !full; i--) {
        const bool in0 = Real(sign) * pq[dir] < h[dir];
        if (in0) {
          pr[0] = pq[0]; pr[1] = pq[1]; pr += 2; nr++;
          if (nr & 8) { full = true; break; }
        }
        Real* nextq = (i > 1) ? pq + 2 : q;
        const bool in1 = Real(sign) * nextq[dir] < h[dir];
        if (in0 != in1) {
          pr[1 - dir] = pq[1 - dir] + (nextq[1 - dir] - pq[1 - dir]) / (nextq[dir] - pq[dir]) * (Real(sign) * h[dir] - pq[dir]);
          pr[dir] = Real(sign) * h[dir];
          pr += 2; nr++;
          if (nr & 8) { full = true; break; }
        }
        pq += 2;
      }
      q = r;
      if (full) { dir = 2; break; }
      r = (q == ret) ? buffer : ret;
      nq = nr;
    }
  }
  if (q != ret) for (int i = 0; i < 2 * nr; i++) ret[i] = q[i];
  return nr;
}

template <class Real>
__device__ __forceinline__ int sp_box_box(const SpatialModel<Real>& Md, SpLds<Real>& S, int sa, int sb, Real* out, V3<Real>& normal,
                                          int& la, int& lb) {
  const Real eps = sizeof(Real) == 4 ? Real(1.1920929e-7) : Real(2.220446049250313e-16);
  la = Md.sh_link[sa]; lb = Md.sh_link[sb];
  V3<Real> u[3], v[3], p1, p2, A, B;
  {
    const Real* La = S.link + la * SP_LINKF;
    const Real* Lb = S.link + lb * SP_LINKF;
    Real Ta[9], Tb[9], ra[9], rb[9];
    for (int k = 0; k < 9; k++) { ra[k] = Md.sh_R[sa][k]; rb[k] = Md.sh_R[sb][k]; }
    mulRR(La + LK_R, ra, Ta);
    mulRR(Lb + LK_R, rb, Tb);
    p1 = ld3(La + LK_P) + mulR(La + LK_R, ld3(Md.sh_p[sa]));
    p2 = ld3(Lb + LK_P) + mulR(Lb + LK_R, ld3(Md.sh_p[sb]));
    for (int j = 0; j < 3; j++) { u[j] = v3<Real>(Ta[j], Ta[3 + j], Ta[6 + j]); v[j] = v3<Real>(Tb[j], Tb[3 + j], Tb[6 + j]); }
    A = ld3(Md.sh_size[sa]) * Real(0.5); B = ld3(Md.sh_size[sb]) * Real(0.5);
  }
  const V3<Real> p = p2 - p1;
  const Real pp[3] = {dot(u[0], p), dot(u[1], p), dot(u[2], p)};
  const Real Av[3] = {A.x, A.y, A.z}, Bv[3] = {B.x, B.y, B.z};
  Real R[3][3], Q[3][3];
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { R[i][j] = dot(u[i], v[j]); Q[i][j] = fabs(R[i][j]); }
  Real s = -inf_<Real>();
  V3<Real> nC = v3<Real>(0, 0, 0);
  int code = 0;
  bool invert = false, sep = false;
  // face axes of box 1, then of box 2
  for (int i = 0; i < 3; i++) {
    const Real e1 = pp[i], s2 = fabs(e1) - (Av[i] + Bv[0] * Q[i][0] + Bv[1] * Q[i][1] + Bv[2] * Q[i][2]);
    sep = sep || (s2 > Real(0));
    if (s2 > s) { s = s2; invert = e1 < Real(0); code = i + 1; }
  }
  for (int j = 0; j < 3; j++) {
    const Real e1 = dot(v[j], p), s2 = fabs(e1) - (Av[0] * Q[0][j] + Av[1] * Q[1][j] + Av[2] * Q[2][j] + Bv[j]);
    sep = sep || (s2 > Real(0));
    if (s2 > s) { s = s2; invert = e1 < Real(0); code = j + 4; }
  }
  if (sep) return 0;
  // edge axes u_i x v_j (i = 0: (0,-R2j,R1j), i = 1: (R2j,0,-R0j), i = 2: (-R1j,R0j,0)), Q padded by 1e-5 like ODE
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Q[i][j] += Real(1.0e-5);
  for (int i = 0; i < 3; i++) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    for (int j = 0; j < 3; j++) {
      const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
      const Real e1 = pp[i2] * R[i1][j] - pp[i1] * R[i2][j];
      Real s2 = fabs(e1) - (Av[i1] * Q[i2][j] + Av[i2] * Q[i1][j] + Bv[j1] * Q[i][j2] + Bv[j2] * Q[i][j1]);
      sep = sep || (s2 > eps);
      Real nv[3] = {0, 0, 0};
      nv[i1] = -R[i2][j]; nv[i2] = R[i1][j];
      const Real l = sqrt(nv[0] * nv[0] + nv[1] * nv[1] + nv[2] * nv[2]);
      if (!sep && l > eps) {
        s2 /= l;
        if (s2 * Real(1.05) > s) { s = s2; nC = v3<Real>(nv[0] / l, nv[1] / l, nv[2] / l); invert = e1 < Real(0); code = 7 + 3 * i + j; }
      }
    }
  }
  if (sep || code == 0) return 0;
  if (code <= 3) normal = u[code - 1];
  else if (code <= 6) normal = v[code - 4];
  else normal = u[0] * nC.x + u[1] * nC.y + u[2] * nC.z;
  if (invert) normal = normal * Real(-1);
  const Real depth = -s;
  if (code > 6) {   // edge-edge
    V3<Real> pa = p1, pb = p2;
    for (int j = 0; j < 3; j++) {
      pa = pa + u[j] * ((dot(normal, u[j]) > Real(0) ? Real(1) : Real(-1)) * Av[j]);
      pb = pb + v[j] * ((dot(normal, v[j]) > Real(0) ? Real(-1) : Real(1)) * Bv[j]);
    }
    const int ia = (code - 7) / 3, ib = (code - 7) % 3;
    const V3<Real> ua = ia == 0 ? u[0] : (ia == 1 ? u[1] : u[2]), ub = ib == 0 ? v[0] : (ib == 1 ? v[1] : v[2]);
    const V3<Real> d3 = pb - pa;
    const Real uaub = dot(ua, ub), q1 = dot(ua, d3), q2 = -dot(ub, d3);
    Real d = Real(1) - uaub * uaub, alpha = Real(0), beta = Real(0);
    if (d > Real(1e-4)) { d = Real(1) / d; alpha = (q1 + uaub * q2) * d; beta = (uaub * q1 + q2) * d; }
    const V3<Real> mid = ((pa + ua * alpha) + (pb + ub * beta)) * Real(0.5);
    out[0] = mid.x; out[1] = mid.y; out[2] = mid.z; out[3] = depth;
    return 1;
  }
  // face case: the reference face belongs to box a (box 1 for codes 1..3, box 2 otherwise)
  const bool first = code <= 3;
  V3<Real> Ra[3], Rb[3];
  for (int j = 0; j < 3; j++) { Ra[j] = first ? u[j] : v[j]; Rb[j] = first ? v[j] : u[j]; }
  const V3<Real> pa = first ? p1 : p2, pb = first ? p2 : p1;
  const Real* Sa = first ? Av : Bv;
  const Real* Sb = first ? Bv : Av;
  const V3<Real> normal2 = first ? normal : normal * Real(-1);
  const Real nr[3] = {dot(Rb[0], normal2), dot(Rb[1], normal2), dot(Rb[2], normal2)};
  const Real anr[3] = {fabs(nr[0]), fabs(nr[1]), fabs(nr[2])};
  int lanr, a1, a2;
  if (anr[1] > anr[0]) { if (anr[1] > anr[2]) { a1 = 0; lanr = 1; a2 = 2; } else { a1 = 0; a2 = 1; lanr = 2; } }
  else { if (anr[0] > anr[2]) { lanr = 0; a1 = 1; a2 = 2; } else { a1 = 0; a2 = 1; lanr = 2; } }
  auto pick = [](const V3<Real>* M3, int k) -> V3<Real> { return k == 0 ? M3[0] : (k == 1 ? M3[1] : M3[2]); };
  auto pickr = [](const Real* a3, int k) -> Real { return k == 0 ? a3[0] : (k == 1 ? a3[1] : a3[2]); };
  const V3<Real> Rbl = pick(Rb, lanr), Rb1 = pick(Rb, a1), Rb2 = pick(Rb, a2);
  const V3<Real> center = pb - pa + Rbl * ((pickr(nr, lanr) < Real(0) ? Real(1) : Real(-1)) * pickr(Sb, lanr));
  const int codeN = first ? code - 1 : code - 4;
  const int code1 = codeN == 0 ? 1 : 0, code2 = codeN == 2 ? 1 : 2;
  const V3<Real> Ra1 = pick(Ra, code1), Ra2 = pick(Ra, code2);
  const Real c1 = dot(center, Ra1), c2 = dot(center, Ra2);
  Real m11 = dot(Ra1, Rb1), m12 = dot(Ra1, Rb2), m21 = dot(Ra2, Rb1), m22 = dot(Ra2, Rb2);
  Real* quad = out + 32;
  Real* ret = out;
  Real* buffer = out + 16;
  {
    const Real k1 = m11 * pickr(Sb, a1), k2 = m21 * pickr(Sb, a1), k3 = m12 * pickr(Sb, a2), k4 = m22 * pickr(Sb, a2);
    quad[0] = c1 - k1 - k3; quad[1] = c2 - k2 - k4; quad[2] = c1 - k1 + k3; quad[3] = c2 - k2 + k4;
    quad[4] = c1 + k1 + k3; quad[5] = c2 + k2 + k4; quad[6] = c1 + k1 - k3; quad[7] = c2 + k2 - k4;
  }
  const Real rect[2] = {pickr(Sa, code1), pickr(Sa, code2)};
  const int nq = sp_clip_rect_quad<Real>(rect, quad, ret, buffer);
  if (nq < 1) return 0;
  Real rx[8], ry[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { rx[j] = j < nq ? ret[2 * j] : Real(0); ry[j] = j < nq ? ret[2 * j + 1] : Real(0); }
  const Real det1 = Real(1) / (m11 * m22 - m12 * m21);
  m11 *= det1; m12 *= det1; m21 *= det1; m22 *= det1;
  const Real SaN = pickr(Sa, codeN);
  int cnum = 0;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    if (j < nq) {
      const Real k1 = m22 * (rx[j] - c1) - m12 * (ry[j] - c2), k2 = -m21 * (rx[j] - c1) + m11 * (ry[j] - c2);
      const V3<Real> pt = center + Rb1 * k1 + Rb2 * k2;
      const Real dep = SaN - dot(normal2, pt);
      if (dep >= Real(0)) {
        const V3<Real> pos = first ? pt + pa : pt + pa - normal * dep;
        out[4 * cnum + 0] = pos.x; out[4 * cnum + 1] = pos.y; out[4 * cnum + 2] = pos.z; out[4 * cnum + 3] = dep;
        cnum++;
      }
    }
  }
  return cnum;
}

}  // namespace dartk
