#!/usr/bin/env python3
"""Generate tests/gpu_kernels/poison_harness.hip (the register-poisoning kernel names every VGPR / AGPR explicitly)."""
import os
regs_v = "\n".join('    asm volatile("v_mov_b32 v%d, %%0" :: "v"(p) : "v%d");' % (i, i) for i in range(8, 256))
regs_a = "\n".join('    asm volatile("v_accvgpr_write_b32 a%d, %%0" :: "v"(p) : "a%d");' % (i, i) for i in range(0, 256))
# subset kernel: register i gets `p` when bit i of the mask is set (VGPRs: bits 0..255 of mv, AGPRs: of ma), `b` otherwise
sub_v = "\n".join('  { const uint32_t x = ((m.v[%d] >> %d) & 1u) ? p : b; asm volatile("v_mov_b32 v%d, %%0" :: "v"(x) : "v%d"); }' % (i // 32, i % 32, i, i) for i in range(8, 256))
sub_a = "\n".join('  { const uint32_t x = ((m.a[%d] >> %d) & 1u) ? p : b; asm volatile("v_accvgpr_write_b32 a%d, %%0" :: "v"(x) : "a%d"); }' % (i // 32, i % 32, i, i) for i in range(0, 256))
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "poison_harness.hip.in")).read().replace("@REGS_V@", regs_v).replace("@REGS_A@", regs_a).replace("@SUB_V@", sub_v).replace("@SUB_A@", sub_a)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "gpu_kernels", "poison_harness.hip")
open(out, "w").write(src)
print("wrote", os.path.normpath(out))
