"""Developer tool: ISA-level experiments on ONE kernel of a hipcc `-S --cuda-device-only` listing.  Applies a named patch to the lines of the
kernel `sym` only, assembles the whole listing into a code object (clang -x assembler + ld.lld) that tools/isa_probe/run_hsaco.py loads with
hipModuleLoad.  usage: python patch_isa.py <in.s> <out.hsaco> <patch-name>"""
import re, subprocess, sys
SYM = "_ZN12_GLOBAL__N_111attn_kernelILi16ELi2ELb0ELb0ELb1EEEvNS_8AttnArgsE"
NOP = "\ts_nop 7\n"


def patch(lines, name):
    out = []
    n = 0
    region = [False]
    for i, l in enumerate(lines):
        ins = l.strip()
        nxt = lines[i + 1].strip() if i + 1 < len(lines) else ""
        if name == "after_swap" and ins.startswith("v_permlane32_swap"):
            out += [l, NOP]; n += 1; continue
        if name == "before_swap" and ins.startswith("v_permlane32_swap"):
            out += [NOP, l]; n += 1; continue
        if name == "after_stage_wait" and ins == "s_waitcnt vmcnt(0)" and nxt.startswith("ds_write_b128"):
            out += [l, NOP]; n += 1; continue
        if name == "after_stage_write" and ins.startswith("ds_write_b128"):
            out += [l, NOP]; n += 1; continue
        if name == "before_stage_wait" and ins == "s_waitcnt vmcnt(0)" and nxt.startswith("ds_write_b128"):
            out += [NOP, l]; n += 1; continue
        if name == "no_prio" and ins == "s_setprio 2":
            out += ["\ts_setprio 0\n"]; n += 1; continue
        if name == "after_gload" and ins.startswith("global_load_dwordx4 v[106:109]"):
            out += [l, NOP]; n += 1; continue
        if name == "before_barrier" and ins == "s_barrier":
            out += [NOP, l]; n += 1; continue
        if name == "after_barrier" and ins == "s_barrier":
            out += [l, NOP]; n += 1; continue
        if name == "after_mfma" and ins.startswith("v_mfma"):
            out += [l, NOP]; n += 1; continue
        if name == "after_exp" and ins.startswith("v_exp_f32"):
            out += [l, "\ts_nop 1\n"]; n += 1; continue
        if name == "after_pk" and ins.startswith("v_pk_"):
            out += [l, "\ts_nop 1\n"]; n += 1; continue
        if name == "after_tr_read_wait" and ins.startswith("s_waitcnt lgkmcnt"):
            out += [l, NOP]; n += 1; continue
        # ---- round-4 targeted probes on the u = 1 rescale / K-store-address registers of the stamped d = 16 build
        if name == "exp1_nop" and ins == "v_exp_f32_e32 v128, v128":
            out += [l, NOP]; n += 1; continue
        if name == "exp1_rename" and (ins == "v_exp_f32_e32 v128, v128" or (ins.startswith("v_pk_mul_f32") and "v[128:129]" in ins)):
            out += [l.replace("v_exp_f32_e32 v128, v128", "v_exp_f32_e32 v132, v128").replace("v[128:129]", "v[132:133]")]; n += 1; continue
        if name == "before_fma1" and ins.startswith("v_fma_f32 v128, "):
            out += [NOP, l]; n += 1; continue
        if name == "kaddr_rename" and (ins.startswith("v_add_u32_e32 v128, ") or ins == "ds_write_b128 v128, v[106:109]"):
            out += [l.replace("v128", "v132")]; n += 1; continue
        if name == "after_setprio0" and ins == "s_setprio 0":
            out += [l, NOP]; n += 1; continue
        if name == "before_kaddr" and ins.startswith("v_add_u32_e32 v128, "):
            out += [NOP, l]; n += 1; continue
        if name == "after_kaddr" and ins.startswith("v_add_u32_e32 v128, "):
            out += [l, NOP]; n += 1; continue
        if name == "loop_top" and "Inner Loop Header" in l and "LBB16_36" in l:
            out += [l, NOP]; n += 1; continue
        if name == "after_cmp" and ins.startswith("v_cmp_neq_f32_e32 vcc"):
            out += [l, NOP]; n += 1; continue
        # ---- prologue probes: the Q fragment loads (v[102:105] = block 0, v[98:101] = block 1, both addressed through v[2:3]) and the
        #      first staging block that re-uses v[2:3] as the K(0) / V(0) address (executed by the staging waves only)
        if name == "wait_after_q1" and ins.startswith("global_load_dwordx4 v[98:101]"):
            out += [l, "\ts_waitcnt vmcnt(0)\n"]; n += 1; continue
        if name == "nop_after_q1" and ins.startswith("global_load_dwordx4 v[98:101]"):
            out += [l] + [NOP] * 8; n += 1; continue
        if name == "wait_after_q0" and ins.startswith("global_load_dwordx4 v[102:105]"):
            out += [l, "\ts_waitcnt vmcnt(0)\n"]; n += 1; continue
        if name == "stage0_rename":
            if ins.startswith("; %bb.17:"):
                region[0] = True
            elif l.startswith(".LBB16_18:"):
                region[0] = False
            if region[0] and ("v2" in ins or "v3" in ins or "v[2:3]" in ins):
                l2 = re.sub(r"\bv2\b", "v8", l); l2 = re.sub(r"\bv3\b", "v9", l2); l2 = l2.replace("v[2:3]", "v[8:9]")
                out += [l2]; n += 1; continue
        if name == "kst0_rename" and (ins == "v_mad_u32_u24 v2, v122, 48, s3" or ins == "ds_write_b128 v2, v[106:109]"):
            out += [l.replace("v2,", "v8,")]; n += 1; continue
        if name == "wait_bb17" and ins.startswith("; %bb.17:"):
            out += [l, "\ts_waitcnt vmcnt(0)\n"]; n += 1; continue
        if name == "wait_bb19" and ins.startswith("; %bb.19:"):
            out += [l, "\ts_waitcnt vmcnt(0)\n"]; n += 1; continue
        if name == "wait_before_zbar" and ins == "s_and_b32 s33, s28, 3":
            out += [l, "\ts_waitcnt vmcnt(0)\n"]; n += 1; continue
        # ---- MFMA whose destination overlaps its own A operand (block 1 of tile 0: v[4:19] <- v[4:7], v[20:35] <- v[20:23]; loop: v[82:97] <- v[82:85])
        def copy_a(dst0, src0):
            return ["\tv_mov_b32_e32 v%d, v%d\n" % (dst0 + i, src0 + i) for i in range(4)] + ["\ts_nop 3\n"]
        if name in ("t0_nooverlap", "all_nooverlap") and ins == "v_mfma_f32_32x32x16_bf16 v[4:19], v[4:7], v[98:101], 0":
            out += copy_a(84, 4) + [l.replace("v[4:7]", "v[84:87]")]; n += 1; continue
        if name in ("t0_nooverlap", "all_nooverlap") and ins == "v_mfma_f32_32x32x16_bf16 v[20:35], v[20:23], v[98:101], 0":
            out += copy_a(88, 20) + [l.replace("v[20:23]", "v[88:91]")]; n += 1; continue
        if name in ("loop_nooverlap", "all_nooverlap") and ins == "v_mfma_f32_32x32x16_bf16 v[82:97], v[82:85], v[98:101], 0":
            out += copy_a(140, 82) + [l.replace("v[82:85],", "v[140:143],")]; n += 1; continue
        if name == "t0_pad_only" and ins in ("v_mfma_f32_32x32x16_bf16 v[4:19], v[4:7], v[98:101], 0", "v_mfma_f32_32x32x16_bf16 v[20:35], v[20:23], v[98:101], 0"):
            out += [NOP, l]; n += 1; continue
        if name == "wait_before_mfma0" and ins == "v_mfma_f32_32x32x16_bf16 v[36:51], v[4:7], v[102:105], 0":
            out += ["\ts_waitcnt vmcnt(0)\n", l]; n += 1; continue
        if name == "nop_before_mfma0" and ins == "v_mfma_f32_32x32x16_bf16 v[36:51], v[4:7], v[102:105], 0":
            out += [NOP] * 4 + [l]; n += 1; continue
        # ---- dump the block-1 scores of tile 0 (M2 -> v[4:19], M4 -> v[20:35]) of waves 0/1 of the first 112 workgroups into the
        #      stamp buffer at +256 KB, before anything modifies them (placed in front of the V(0) staging branch)
        if name == "dump_s1" and ins == "s_cbranch_vccnz .LBB16_26":
            code = ["s_cmp_lt_u32 s28, 2", "s_cbranch_scc0 .Ldump_skip", "s_cmp_lt_u32 s2, 112", "s_cbranch_scc0 .Ldump_skip",
                    "s_getpc_b64 s[60:61]", "s_add_u32 s60, s60, dc_attn_stamp_buf@rel32@lo+4", "s_addc_u32 s61, s61, dc_attn_stamp_buf@rel32@hi+12",
                    "s_lshl_b32 s62, s2, 1", "s_add_u32 s62, s62, s28", "s_lshl_b32 s62, s62, 13", "s_add_u32 s62, s62, 0x40000",
                    "s_add_u32 s60, s60, s62", "s_addc_u32 s61, s61, 0", "v_lshlrev_b32_e32 v84, 2, v122"]
            code += ["global_store_dword v84, v%d, s[60:61] offset:%d" % (4 + r, 256 * r) for r in range(16)]
            code += ["s_add_u32 s60, s60, 4096", "s_addc_u32 s61, s61, 0"]
            code += ["global_store_dword v84, v%d, s[60:61] offset:%d" % (20 + r, 256 * r) for r in range(16)]
            out += ["\t" + c + "\n" for c in code] + [".Ldump_skip:\n", l]; n += 1; continue
        # ---- dump block 1's tile-0 probabilities (pf[1]: v[60:63], v[64:67], v[56:59], v[52:55]), m_run1 (v128), m_run0 (v120) at the loop
        #      pre-header (waves 0/1, first 112 workgroups) -> stamp buffer + 256 KB
        if name == "dump_p1" and l.startswith(".LBB16_36:"):
            regs = [60, 61, 62, 63, 64, 65, 66, 67, 56, 57, 58, 59, 52, 53, 54, 55, 128, 120]
            code = ["s_cmp_lt_u32 s28, 2", "s_cbranch_scc0 .Ldump_skip", "s_cmp_lt_u32 s2, 112", "s_cbranch_scc0 .Ldump_skip",
                    "s_getpc_b64 s[60:61]", "s_add_u32 s60, s60, dc_attn_stamp_buf@rel32@lo+4", "s_addc_u32 s61, s61, dc_attn_stamp_buf@rel32@hi+12",
                    "s_lshl_b32 s62, s2, 1", "s_add_u32 s62, s62, s28", "s_lshl_b32 s62, s62, 13", "s_add_u32 s62, s62, 0x40000",
                    "s_add_u32 s60, s60, s62", "s_addc_u32 s61, s61, 0", "v_lshlrev_b32_e32 v84, 2, v122"]
            for i, r in enumerate(regs):
                if i == 16:
                    code += ["s_add_u32 s60, s60, 4096", "s_addc_u32 s61, s61, 0"]
                code += ["global_store_dword v84, v%d, s[60:61] offset:%d" % (r, 256 * (i % 16))]
            out += ["\t" + c + "\n" for c in code] + [".Ldump_skip:\n", l]; n += 1; continue
        # ---- the instruction whose result is wrong (dump_p1): block 1's last packed FMA of the tile-0 softmax
        m = re.match(r"v_pk_fma_f32 v\[4:5\], s\[36:37\], v\[(\d+):(\d+)\], v\[54:55\] ", ins)
        if m and (name == "all_scalar_fma" or (name == "victim_scalar_fma" and m.group(1) == "34")):
            out += ["\tv_fma_f32 v4, s36, v%s, -v55\n" % m.group(1), "\tv_fma_f32 v5, s36, v%s, -v55\n" % m.group(2)]; n += 1; continue
        if name == "pad_before_v0_store" and ins.startswith("; %bb.25:"):
            out += [l] + [NOP] * 8; n += 1; continue
        if name == "pad_after_v0_store" and ins == "ds_write_b128 v69, v[106:109] offset:10240":
            out += [l] + [NOP] * 8; n += 1; continue
        if name == "exp3_after" and ins == "v_exp_f32_e32 v3, v3" and lines[i - 1].strip().startswith("v_pk_fma_f32 v[4:5], s[36:37], v[34:35]"):
            n += 1; continue                                   # dropped here ...
        if name == "exp3_after" and ins == "v_exp_f32_e32 v5, v5" and lines[i - 1].strip() == "v_exp_f32_e32 v4, v4" and lines[i - 2].strip().startswith("s_cmp_eq_u64 s[8:9]"):
            out += [l, "\tv_exp_f32_e32 v3, v3\n"]; n += 1; continue   # ... re-inserted behind the dependent pair
        if name == "nop_after_victim" and ins.startswith("v_pk_fma_f32 v[4:5], s[36:37], v[34:35]"):
            out += [l, "\ts_nop 0\n"]; n += 1; continue
        if name in ("shift4", "shift8", "shift12", "shift16") and ins == "s_mov_b32 s3, 0xff800000":
            out += [l] + ["\ts_nop 0\n"] * (int(name[5:]) // 4); n += 1; continue
        out.append(l)
    return out, n


src, dst, name = sys.argv[1:4]
L = open(src).readlines()
a = next(i for i, l in enumerate(L) if l.startswith(SYM + ":"))
b = next(i for i in range(a, len(L)) if L[i].strip() == "s_endpgm")
body, n = patch(L[a:b + 1], name) if name != "none" else (L[a:b + 1], 0)
tmp = dst.replace(".hsaco", ".s")
rest = L[b + 1:]
if name in ("dump_s1", "dump_p1"):                    # the dumps use s[60:62]
    k = next(i for i, l in enumerate(rest) if l.strip() == ".amdhsa_kernel " + SYM)
    m = next(i for i in range(k, len(rest)) if ".amdhsa_next_free_sgpr" in rest[i])
    rest[m] = "\t\t.amdhsa_next_free_sgpr 64\n"
open(tmp, "w").writelines(L[:a] + body + rest)
obj = dst.replace(".hsaco", ".o")
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", tmp, "-o", obj])
subprocess.check_call(["/opt/rocm/lib/llvm/bin/ld.lld", "-shared", obj, "-o", dst])
print(f"{name}: {n} sites patched -> {dst}")
