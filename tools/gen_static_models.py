#!/usr/bin/env python3
"""Generate dart_env_amd/csrc/static_models.hpp: the shipped model cards baked in as compile-time constants.

The step kernel is templated on its parameter block.  With the runtime block (`Params<Real,T>`, ~110 scalars) the
compiler keeps everything in SGPRs, spills some to VGPR lanes and cannot drop terms that are zero for a given model.
For the env ids this package ships, this script restates dart_stepper.hip's `fill_params` in Python and prints the
values as `static constexpr` members, so the specialised kernel folds them into immediates (and deletes the zero
terms).  The library only selects a specialised kernel when the runtime block built from the caller's card is
bit-identical to the baked one (`matches()`); any modified card falls back to the generic kernel.
"""
import math
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dart_env_amd.model_card import card_for  # noqa: E402

# struct name -> (env id, every capsule collides?, topology trait, links, candidate capsules, actions, capsule links)
TOPOS = {
    "HopperStatic": ("DartHopper-v1", False, "HopperTopo", 4, 1, 3, [3]),
    "HopperAllStatic": ("DartHopper-v1", True, "HopperAllTopo", 4, 4, 3, [0, 1, 2, 3]),
    "Walker2dStatic": ("DartWalker2d-v1", False, "Walker2dTopo", 7, 2, 6, [3, 6]),
    "Walker2dAllStatic": ("DartWalker2d-v1", True, "Walker2dAllTopo", 7, 7, 6, [0, 1, 2, 3, 4, 5, 6]),
    # round 6: the half cheetah (welded head folded into the torso link, joint springs, eight capsules).  With the runtime block its fp64
    # kernel spilled 226 SGPRs into VGPR lanes and 2.1 KB per lane into scratch; the Walker2d kernels show what baking buys: 1 936 -> 352 B.
    "CheetahStatic": ("DartHalfCheetah-v1", True, "CheetahTopo", 7, 8, 6, [0, 0, 1, 2, 3, 4, 5, 6]),
}
JT_WELD = 0


def lit(x):
    if math.isinf(x):
        return ("-" if x < 0 else "") + "Real(__builtin_huge_val())"
    return "Real(%r)" % float(x)


def params_from_card(c, NL, NC, NA, clinks):
    """Python restatement of fill_params<Real,T> (dart_env_amd/csrc/dart_stepper.hip)."""
    P = {}
    nd = NL + 2
    x0 = sum(c.T_pj[b][3] - c.T_cj[b][3] for b in range(3))
    y0 = sum(c.T_pj[b][7] - c.T_cj[b][7] for b in range(3))
    P["root_x0"], P["root_y0"] = x0, y0
    for name in ("sigma", "mass", "cx", "cy", "izz", "jx", "jy", "lo", "hi"):
        P[name] = [0.0] * NL
    # bodies -> links: a welded body shares its parent's link, shifted by the weld offset (fill_params: same expressions, same order --
    # that function evaluates them without fp contraction so that the doubles printed here are the ones it computes)
    link_of_body, body_of_link = {}, []
    wx, wy = [0.0] * c.nbodies, [0.0] * c.nbodies
    for b in range(2, c.nbodies):
        if c.jtype[b] == JT_WELD:
            pb = c.parent[b]
            link_of_body[b] = link_of_body[pb]
            wx[b] = wx[pb] + c.T_pj[b][3] - c.T_cj[b][3]
            wy[b] = wy[pb] + c.T_pj[b][7] - c.T_cj[b][7]
        else:
            link_of_body[b] = len(body_of_link)
            body_of_link.append(b)
    assert len(body_of_link) == NL
    for k in range(NL):
        b = body_of_link[k]
        P["jx"][k] = c.T_pj[b][3] if k > 0 else 0.0
        P["jy"][k] = c.T_pj[b][7] if k > 0 else 0.0
        P["sigma"][k] = 1.0 if c.axes[b][2] > 0 else -1.0
        P["mass"][k], P["cx"][k], P["cy"][k], P["izz"][k] = c.mass[b], c.com[b][0], c.com[b][1], c.inertia[b][8]
        d = 2 + k
        assert c.dof_offset[b] == d
        lim = c.limited[d] != 0
        P["lo"][k] = c.lower[d] if lim else -math.inf
        P["hi"][k] = c.upper[d] if lim else math.inf
    for b in range(2, c.nbodies):   # fold the welded bodies in: composite mass, COM, inertia about the new COM
        if c.jtype[b] != JT_WELD or c.mass[b] == 0:
            continue
        k = link_of_body[b]
        lm, lcx, lcy, lizz = P["mass"][k], P["cx"][k], P["cy"][k], P["izz"][k]
        mb, bx, by = c.mass[b], wx[b] + c.com[b][0], wy[b] + c.com[b][1]
        m = lm + mb
        nx, ny = (lm * lcx + mb * bx) / m, (lm * lcy + mb * by) / m
        lizz = lizz + lm * ((lcx - nx) * (lcx - nx) + (lcy - ny) * (lcy - ny)) + c.inertia[b][8] + mb * ((bx - nx) * (bx - nx) + (by - ny) * (by - ny))
        P["mass"][k], P["cx"][k], P["cy"][k], P["izz"][k] = m, nx, ny, lizz
    P["damp"] = [c.damping[d] for d in range(nd)]
    P["stiff"] = [c.stiffness[d] for d in range(nd)]
    P["rest"] = [c.rest[d] for d in range(nd)]
    P["sqe"] = [math.sqrt(c.dt * c.damping[d] + c.dt * c.dt * c.stiffness[d]) for d in range(nd)]   # derived: not part of matches()
    P["q0"] = [c.init_pos[d] for d in range(nd)]
    P["dq0"] = [c.init_vel[d] for d in range(nd)]
    for name in ("e1x", "e1y", "e2x", "e2y", "rad"):
        P[name] = []
    for s in range(c.nshapes):
        if not c.shape_collidable[s]:
            continue
        S = c.shape_pose[s]
        hl = 0.5 * c.shape_size[s][1]
        sb = c.shape_body[s]
        assert link_of_body[sb] == clinks[len(P["rad"])]
        P["e1x"].append(wx[sb] + S[3] + hl * S[2]); P["e1y"].append(wy[sb] + S[7] + hl * S[6])
        P["e2x"].append(wx[sb] + S[3] - hl * S[2]); P["e2y"].append(wy[sb] + S[7] - hl * S[6])
        P["rad"].append(c.shape_size[s][0])
    assert len(P["rad"]) == NC
    P["dt"], P["ground_y"], P["g"], P["mu"] = c.dt, c.ground_y, -c.gravity[1], c.friction
    P["erp_dt"], P["max_erv"], P["limit_erp_dt"], P["cfm1"] = c.erp / c.dt, c.max_erv, c.limit_erp / c.dt, 1.0 + c.cfm
    P["ccfm1"] = 1.0 + c.contact_cfm
    P["act_scale"] = [c.act_scale[k] for k in range(NA)]
    P["act_lo"] = [c.act_low[k] for k in range(NA)]
    P["act_hi"] = [c.act_high[k] for k in range(NA)]
    P["alive"], P["ctrl_cost"], P["pen_each"] = c.alive_bonus, c.ctrl_cost, c.limit_penalty * 1.5
    P["pen_margin"], P["h_lo"], P["h_hi"], P["ang_max"] = c.penalty_margin, c.height_lo, c.height_hi, c.angle_max
    P["s_max"], P["v_clip"], P["inv_envdt"], P["noise"] = c.state_abs_max, c.obs_vel_clip, 1.0 / (c.dt * c.frame_skip), c.reset_noise
    P["noise_v"] = c.reset_noise_vel
    ints = dict(frame_skip=c.frame_skip, task=c.task, penalty_link=(c.penalty_dof - 2 if c.penalty_dof >= 2 else -1),
                impulse_M=(1 if c.impulse_inertia == 0 else 0))
    ints["cbody"] = [c.shape_body[s] for s in range(c.nshapes) if c.shape_collidable[s]]
    return P, ints


def emit(name):
    env_id, every, topo, NL, NC, NA, clinks = TOPOS[name]
    c = card_for(env_id, all_bodies_collide=every)
    P, ints = params_from_card(c, NL, NC, NA, clinks)
    out = ["// %s%s  (generated by tools/gen_static_models.py from dart_env_amd/models + model_card.py; do not edit)" % (env_id, ", every capsule collides" if every else ", feet only"),
           "template <class Real>", "struct %s {" % name, "  using Topo = %s;" % topo,
           "  static constexpr bool is_static = true;"]
    match = []
    for k, v in P.items():
        if isinstance(v, list):
            out.append("  static constexpr Real %s[%d] = {%s};" % (k, len(v), ", ".join(lit(x) for x in v)))
            for i in range(len(v)):
                if k != "sqe":
                    match.append("R.%s[%d] == %s[%d]" % (k, i, k, i))
        else:
            out.append("  static constexpr Real %s = %s;" % (k, lit(v)))
            match.append("R.%s == %s" % (k, k))
    for k, v in ints.items():
        if isinstance(v, list):
            out.append("  static constexpr int %s[%d] = {%s};" % (k, len(v), ", ".join(str(x) for x in v)))
            match += ["R.%s[%d] == %s[%d]" % (k, i, k, i) for i in range(len(v))]
            continue
        out.append("  static constexpr int %s = %d;" % (k, v))
        match.append("R.%s == %s" % (k, k))
    out += ["  __device__ __host__ static constexpr bool zero(int f, int k) {",
            "    return (f == ZF_jx ? jx[k] : f == ZF_jy ? jy[k] : f == ZF_cx ? cx[k] : f == ZF_cy ? cy[k] : f == ZF_damp ? damp[k] : stiff[k]) == Real(0);",
            "  }",
            "  // runtime members (not part of the model)",
            "  int max_steps, solver, iters1, iters2, force_slow;", "  unsigned long long* stats;", "  Extras<Real> ex;",
            "  static bool matches(const Params<Real, %s>& R) {" % topo,
            "    return " + " &&\n           ".join(match) + ";", "  }", "};", ""]
    return "\n".join(out)


def main():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dart_env_amd", "csrc", "static_models.hpp")
    body = ["// static_models.hpp -- GENERATED by tools/gen_static_models.py; compile-time parameter blocks of the shipped models.",
            "#pragma once", '#include "planar_kernel.hpp"', "", "namespace dartk {", ""]
    for name in TOPOS:
        body.append(emit(name))
    body.append("}  // namespace dartk")
    text = "\n".join(body) + "\n"
    if not os.path.exists(path) or open(path).read() != text:
        with open(path, "w") as f:
            f.write(text)
        print("wrote", path)


if __name__ == "__main__":
    main()
