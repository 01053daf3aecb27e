"""Generates megapath_amd/csrc/kernels_seed_x2.hip -- seed_kernel with two (read, strand) items per lane -- from the text of seed_kernel in
kernels_seed.hip: the item assignment, the issue phase and the consume phase are each wrapped in a loop over the lane's two items, with
references named as the single-item kernel names its variables.  python scripts/make_seed_x2.py"""
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(ROOT, "megapath_amd", "csrc", "kernels_seed.hip")).read()
a = src.index("template <int MINW>      // waves per SIMD")
b = src.index("// Items with more than MP_SEED_SLAB seeds")
text = src[a:b]
i_decl = text.index("    mp_lane_state s;")
i_pool = text.index("    // Work queue: one global counter")
i_loop = text.index("    for (;;) {\n        wave_trips++;")
i_need = text.index("        const unsigned long long need = __ballot(!have);")
i_any = text.index("        if (!__any(have)) {")
i_issue_if = text.index("        if (have) {\n            lane_busy++;")
i_wait_cmt = text.index("        // the one wait of the trip")
i_consume_cmt = text.index("        // ---------------------------------------------------------------- consume")
i_consume_if = text.index("        if (have) {\n            const uint32_t kind = s.kind;\n            uint32_t next = MK_NONE;")
i_pop = text.index("#pragma clang diagnostic pop")
i_after_loop = text.index("    { uint32_t lb = lane_busy;")
head, decl, pool = text[:i_decl], text[i_decl:i_pool], text[i_pool:i_loop]
assign_block, issue_block = text[i_need:i_any], text[i_issue_if:i_wait_cmt]
wait_block, consume_block, tail = text[i_wait_cmt:i_consume_cmt], text[i_consume_if:i_pop], text[i_after_loop:]


def indent(blk, n=4):
    return "\n".join((" " * n + l if l.strip() else l) for l in blk.split("\n"))


out = []
h = head[head.index("seed_kernel("):]
h = ("// seed_kernel with TWO (read, strand) items per lane: both items' loads are issued before the one wait of the trip, then both are consumed\n"
     "// (review item 5a of round 2; measurement variant, MP_SEED_ITEMS=2).  The body of each phase is seed_kernel's, applied to item j of the lane.\n"
     "__global__ void __launch_bounds__(256)\nseed_kernel_x2(" + h[len("seed_kernel("):])
h = h.replace("    uint32_t *const sl = s_scan + threadIdx.x * (S | 1u);\n",
              "    uint32_t *const sl2[2] = {s_scan + (threadIdx.x * 2u) * (S | 1u), s_scan + (threadIdx.x * 2u + 1u) * (S | 1u)};\n")
out.append(h)
d = decl.replace("    mp_lane_state s;\n", "    mp_lane_state st[2];\n    bool hv[2] = {false, false};\n#pragma unroll\n    for (int j = 0; j < 2; j++) {\n        mp_lane_state &s = st[j];\n")
d = d.replace("    bool have = false;\n", "")
nd = [("    " + l if l.startswith("    s.") else l) for l in d.split("\n")]
nd.insert(max(i for i, l in enumerate(nd) if l.startswith("        s.")) + 1, "    }")
out.append("\n".join(nd))
out.append(pool)
out.append("    for (;;) {\n        wave_trips++;\n")
out.append("#pragma unroll\n        for (int j = 0; j < 2; j++) {\n            mp_lane_state &s = st[j];\n            bool &have = hv[j];\n" + indent(assign_block) + "        }\n")
out.append("        if (!__any(hv[0] || hv[1])) { if (drained && pool_next == pool_end) break; else continue; }   // nothing to do this trip\n")
out.append("        uint4 Q0[2], Q1[2], Q2[2], Q3[2];\n        uint32_t A0[2], A1[2], B0[2], B1[2], ALEN[2];\n#pragma unroll\n        for (int j = 0; j < 2; j++) "
           "asm volatile(\"\" : \"=v\"(Q0[j]), \"=v\"(Q1[j]), \"=v\"(Q2[j]), \"=v\"(Q3[j]), \"=v\"(A0[j]), \"=v\"(A1[j]), \"=v\"(B0[j]), \"=v\"(B1[j]), \"=v\"(ALEN[j]));\n")
out.append("#pragma clang diagnostic push\n#pragma clang diagnostic ignored \"-Wuninitialized\"\n#pragma clang diagnostic ignored \"-Wsometimes-uninitialized\"\n"
           "#pragma clang diagnostic ignored \"-Wconditional-uninitialized\"\n")
refs = ("            mp_lane_state &s = st[j];\n            bool &have = hv[j];\n            uint32_t *const sl = sl2[j];\n"
        "            uint4 &q0 = Q0[j], &q1 = Q1[j], &q2 = Q2[j], &q3 = Q3[j];\n            uint32_t &a0 = A0[j], &a1 = A1[j], &b0 = B0[j], &b1 = B1[j], &alen = ALEN[j];\n"
        "            (void)sl; (void)q1; (void)q2; (void)q3; (void)a1; (void)b0; (void)b1; (void)alen;\n")
out.append("#pragma unroll\n        for (int j = 0; j < 2; j++) {\n" + refs + indent(issue_block) + "        }\n")
out.append(wait_block)
out.append("        // ---------------------------------------------------------------- consume\n#pragma unroll\n        for (int j = 0; j < 2; j++) {\n" + refs + indent(consume_block) + "        }\n")
out.append("#pragma clang diagnostic pop\n    }\n")
out.append(tail)
with open(os.path.join(ROOT, "megapath_amd", "csrc", "kernels_seed_x2.hip"), "w") as f:
    f.write("// megapath_amd/csrc/kernels_seed_x2.hip -- product code (measurement variant): GENERATED from seed_kernel's text by scripts/make_seed_x2.py;\n"
            "// see kernels_seed.hip for every comment.  Included by api.hip after kernels_seed.hip.\n#pragma once\n\n" + "".join(out) + "\n")
