// spatial_free_root.hpp -- DART FreeJoint root of the tree kernel (dog.skel): SO(3) exp / log, re-centred rotation chart, body-twist pose update.
// Part of the gfx950 tree kernel; overview in spatial_kernel.hpp, design in DESIGN.md section 4.2.
#pragma once
#include "spatial_model.hpp"

namespace dartk {

// ------------------------------------------------------------------ DART FreeJoint root (dog.skel)
// Public coordinates: q[0:3] rotation vector, q[3:6] translation, dq[0:6] = twist of the child joint frame in that frame;
// DART integrates the pose as Q <- Q * [exp(w dt), v dt].  The dynamics run on the internal chain (translation x y z,
// rotations about x y z in a chart centred on the current orientation); S.root keeps R, p and the twist, the internal
// coordinates are re-derived from them before every world
// step and the new velocities are mapped back with the exact instantaneous Jacobian -- only the parametrisation
// differs from DART, not the integrator.
template <class Real>
__device__ __forceinline__ void sp_so3_exp(V3<Real> r, Real* R) {
  const Real th2 = dot(r, r), th = sqrt(th2);
  Real a, b;
  if (th < Real(1e-4)) { a = Real(1) - th2 / Real(6); b = Real(0.5) - th2 / Real(24); }
  else { Real sn, cs; sincos_<Real>(th, sn, cs); a = sn / th; b = (Real(1) - cs) / th2; }
  const Real K[9] = {0, -r.z, r.y, r.z, 0, -r.x, -r.y, r.x, 0};
  Real K2[9];
  mulRR(K, K, K2);
  for (int k = 0; k < 9; k++) R[k] = ((k % 4 == 0) ? Real(1) : Real(0)) + a * K[k] + b * K2[k];
}
// log map through the unit quaternion (largest-pivot extraction, then 2 atan2(|v|, w)): well conditioned at every angle,
// including rotations by pi where acos(trace) and R - R^T lose half of the digits -- the pose makes this round trip once per
// env step because the public state is DART's rotation vector.
template <class Real>
__device__ __forceinline__ V3<Real> sp_so3_log(const Real* R) {
  const Real tr = R[0] + R[4] + R[8];
  Real w, x, y, z;
  if (tr > Real(0)) {
    const Real s = sqrt(tr + Real(1)) * Real(2);
    w = s * Real(0.25); x = (R[7] - R[5]) / s; y = (R[2] - R[6]) / s; z = (R[3] - R[1]) / s;
  } else if (R[0] > R[4] && R[0] > R[8]) {
    const Real s = sqrt(Real(1) + R[0] - R[4] - R[8]) * Real(2);
    w = (R[7] - R[5]) / s; x = s * Real(0.25); y = (R[1] + R[3]) / s; z = (R[2] + R[6]) / s;
  } else if (R[4] > R[8]) {
    const Real s = sqrt(Real(1) + R[4] - R[0] - R[8]) * Real(2);
    w = (R[2] - R[6]) / s; x = (R[1] + R[3]) / s; y = s * Real(0.25); z = (R[5] + R[7]) / s;
  } else {
    const Real s = sqrt(Real(1) + R[8] - R[0] - R[4]) * Real(2);
    w = (R[3] - R[1]) / s; x = (R[2] + R[6]) / s; y = (R[5] + R[7]) / s; z = s * Real(0.25);
  }
  if (w < Real(0)) { w = -w; x = -x; y = -y; z = -z; }   // angle in [0, pi]
  const Real nv = sqrt(x * x + y * y + z * z);
  const Real k = nv < Real(1e-6) ? Real(2) / w : Real(2) * atan2(nv, w) / nv;
  return v3<Real>(x * k, y * k, z * k);
}
// S.root -> internal coordinates of the six root links (lane 0).  The rotation chart is centred on the current orientation:
// joint rotation = Rx(a) Ry(b) Rz(c) R0 with R0 = S.root[0:9] and a = b = c = 0, so the rates are the angular velocity in
// the joint's parent frame (E = I) and the chart has no singularity however far the body turns; R0 enters the forward pass
// as part of the last root link's joint-to-child transform (sp_forward / sp_kinematics).
template <class Real>
__device__ __forceinline__ void sp_free_root_to_internal(SpLds<Real>& S) {
  const Real* R = S.root;
  const V3<Real> ww = mulR(R, ld3(S.root + 12)), pd = mulR(R, ld3(S.root + 15));
  S.q[0] = Real(0); S.q[1] = Real(0); S.q[2] = Real(0);
  S.q[3] = S.root[9]; S.q[4] = S.root[10]; S.q[5] = S.root[11];
  S.dq[0] = ww.x; S.dq[1] = ww.y; S.dq[2] = ww.z;
  S.dq[3] = pd.x; S.dq[4] = pd.y; S.dq[5] = pd.z;
}
// after the velocity update: new internal rates (at the old pose) -> new body twist; DART's pose update
template <class Real>
__device__ __forceinline__ void sp_free_root_advance(SpLds<Real>& S, Real dt) {
  Real* R = S.root;
  const V3<Real> ww = v3<Real>(S.dq[0], S.dq[1], S.dq[2]);
  const V3<Real> pd = v3<Real>(S.dq[3], S.dq[4], S.dq[5]);
  const V3<Real> wb = v3<Real>(R[0] * ww.x + R[3] * ww.y + R[6] * ww.z, R[1] * ww.x + R[4] * ww.y + R[7] * ww.z, R[2] * ww.x + R[5] * ww.y + R[8] * ww.z);
  const V3<Real> vb = v3<Real>(R[0] * pd.x + R[3] * pd.y + R[6] * pd.z, R[1] * pd.x + R[4] * pd.y + R[7] * pd.z, R[2] * pd.x + R[5] * pd.y + R[8] * pd.z);
  st3(S.root + 12, wb); st3(S.root + 15, vb);
  S.root[9] += dt * pd.x; S.root[10] += dt * pd.y; S.root[11] += dt * pd.z;   // p += R v_b dt = pdot dt
  Real dR[9], Rn[9];
  sp_so3_exp<Real>(wb * dt, dR);
  mulRR(R, dR, Rn);
  for (int k = 0; k < 9; k++) R[k] = Rn[k];
}
// DART integrates the BODY-FRAME twist: twist' = twist + dt twist_acc.  With dq_int = T(q) twist the accelerations map as
// qdd_int = T twist_acc + Tdot twist, so the internal velocity that corresponds to DART's unconstrained one is
// dq_int + dt qdd_int - dt Tdot twist:  rotation (E rates = R w_b): -Tdot twist = E^-1 Edot rates = (rb rc, -ra rc, ra rb) at
// the chart centre;  translation (pdot = R v_b): -Tdot twist = -(w x pdot).  Applied to S.dq[0:6] after the bias forces have
// been computed from the true velocities; every later use of S.dq in the world step is at velocity level.
template <class Real>
__device__ __forceinline__ void sp_free_root_velocity_correction(SpLds<Real>& S, Real dt) {
  const Real ra = S.dq[0], rb = S.dq[1], rc = S.dq[2];
  const V3<Real> wxp = cross(v3<Real>(ra, rb, rc), v3<Real>(S.dq[3], S.dq[4], S.dq[5]));
  S.dq[0] += dt * (rb * rc); S.dq[1] -= dt * (ra * rc); S.dq[2] += dt * (ra * rb);
  S.dq[3] -= dt * wxp.x; S.dq[4] -= dt * wxp.y; S.dq[5] -= dt * wxp.z;
}
// public root coordinates (already in S.q / S.dq[0:6]) -> S.root
template <class Real>
__device__ __forceinline__ void sp_free_root_load(SpLds<Real>& S) {
  sp_so3_exp<Real>(v3<Real>(S.q[0], S.q[1], S.q[2]), S.root);
  for (int k = 0; k < 3; k++) { S.root[9 + k] = S.q[3 + k]; S.root[12 + k] = S.dq[k]; S.root[15 + k] = S.dq[3 + k]; }
}
// S.root -> public coordinates in S.q / S.dq[0:6]
template <class Real>
__device__ __forceinline__ void sp_free_root_store(SpLds<Real>& S) {
  const V3<Real> r = sp_so3_log<Real>(S.root);
  S.q[0] = r.x; S.q[1] = r.y; S.q[2] = r.z;
  for (int k = 0; k < 3; k++) { S.q[3 + k] = S.root[9 + k]; S.dq[k] = S.root[12 + k]; S.dq[3 + k] = S.root[15 + k]; }
}

}  // namespace dartk
