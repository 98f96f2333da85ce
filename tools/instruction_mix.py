"""Static instruction mix of the compiled kernels of one csrc/*.hip file (gfx950 assembly from hipcc -S): vector ALU, DPP, matrix,
LDS, vector-memory, scalar, waits, barriers per kernel.  Not a profile -- loops count once, skipped rounds count fully -- but enough
to see what a kernel is made of: the GAT layer's pullback was 2 367 + 560 (DPP) vector instructions against 48 matrix instructions
per wave and tile before DESIGN 5.11's rewrite, 1 220 + 112 against 80 after.
usage: python tools/instruction_mix.py neuralgraphpde.jl_amd/csrc/gat_fused.hip [substring of the kernel names]"""
import collections, os, re, subprocess, sys, tempfile

src = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
out = os.path.join(tempfile.mkdtemp(), "k.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-mllvm",
                "-amdgpu-kernarg-preload-count=16", "-I", os.path.dirname(src), "-S", "--cuda-device-only", "-o", out, src],
               check=True, stderr=subprocess.DEVNULL)
txt = open(out).read()
for name in re.findall(r"^\s*\.amdhsa_kernel (\S+)", txt, re.M):
    if want not in name:
        continue
    i = txt.index("\n" + name + ":")
    body = txt[i:txt.index(".Lfunc_end", i)]
    c = collections.Counter()
    for line in body.split("\n"):
        line = line.strip()
        m = re.match(r"([a-z_0-9]+)", line)
        if not m or line.startswith((".", ";", "//")) or line.endswith(":"):
            continue
        op = m.group(1)
        if op.startswith("v_mfma"): c["matrix"] += 1
        elif op.startswith("v_") and "dpp" in line: c["valu_dpp"] += 1
        elif op.startswith("v_"): c["valu"] += 1
        elif op.startswith("ds_"): c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "scratch_", "flat_")): c["vmem"] += 1
        elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
        elif op.startswith("s_barrier"): c["barrier"] += 1
        elif op.startswith("s_"): c["scalar"] += 1
    meta = txt[txt.index(".amdhsa_kernel " + name):]
    vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", meta).group(1)
    lds = re.search(r"\.amdhsa_group_segment_fixed_size (\d+)", meta).group(1)
    sc = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", meta).group(1)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::|ngpde::|void ", "", dem)
    dem = re.sub(r"\(.*", "", dem)
    print(f"{dem}: " + ", ".join(f"{k} {c[k]}" for k in ("valu", "valu_dpp", "matrix", "lds", "vmem", "scalar", "waitcnt", "barrier")) +
          f" | vgpr {vg}, lds {lds} B, scratch {sc} B")
