"""Register / LDS / scratch usage of the kernels of one .hip file (build container, no GPU):
python tools/kres.py whisper.tflite_amd/csrc/k_gemm_planes.hip [name-filter]
Compiles with -save-temps into /tmp/kres and reads the .amdhsa_kernel blocks of the gfx950 assembly; also counts the
MFMA / ds_read / barrier instructions of each kernel body so a schedule edit can be checked without a GPU."""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.abspath(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = "/tmp/kres"
os.makedirs(out, exist_ok=True)
stem = os.path.splitext(os.path.basename(src))[0]
cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", f"-I{root}/include", f"-I{root}/whisper.tflite_amd/csrc",
       "--offload-arch=gfx950", "-ffp-contract=fast", "-save-temps", "-c", src, "-o", f"{out}/{stem}.o"]
r = subprocess.run(cmd, cwd=out, capture_output=True, text=True)
if r.returncode:
    sys.stderr.write(r.stderr)
    sys.exit(r.returncode)
asm = open(f"{out}/{stem}-hip-amdgcn-amd-amdhsa-gfx950.s").read()
demangle = lambda n: subprocess.run(["/usr/bin/c++filt", n], capture_output=True, text=True).stdout.strip()
for blk in asm.split(".amdhsa_kernel ")[1:]:
    name = blk.split("\n")[0].strip()
    dn = demangle(name)
    if flt and flt not in dn:
        continue
    g = lambda k: (re.search(r"\.amdhsa_" + k + r"\s+(\S+)", blk) or [None, "?"])[1]
    body = asm[asm.index(name + ":"):]
    body = body[:body.index("s_endpgm")]
    cnt = lambda pat: len(re.findall(pat, body))
    spill = re.search(re.escape(name) + r".*?\.vgpr_spill_count:\s+(\d+)", asm[asm.index("amdhsa.kernels"):], re.S)
    print(f"{dn[:110]}\n    vgpr+agpr {g('next_free_vgpr')} (accum_offset {g('accum_offset')}) sgpr {g('next_free_sgpr')} "
          f"scratch {g('private_segment_fixed_size')} spill {spill.group(1) if spill else '?'} | mfma {cnt(r'v_mfma')} "
          f"ds_read {cnt(r'ds_read')} ds_write {cnt(r'ds_write')} barrier {cnt(r's_barrier')} glds {cnt(r'global_load_lds')} "
          f"accvgpr_mov {cnt(r'v_accvgpr')}")


def sequence(body):
    """Run-length summary of a kernel body's instruction stream (opcode xN), one line per barrier-delimited phase."""
    ops = []
    for line in body.split("\n"):
        t = line.strip()
        if not t or t[0] in ";." or t.endswith(":"):
            continue
        ops.append(t.split()[0])
    phase, prev, n = [], None, 0
    for op in ops + [None]:
        if op == prev:
            n += 1
            continue
        if prev:
            phase.append(f"{prev}x{n}" if n > 1 else prev)
        if prev == "s_barrier":
            print("   ", " ".join(phase))
            phase = []
        prev, n = op, 1
    print("   ", " ".join(phase))


if os.environ.get("KRES_SEQ"):
    for blk in asm.split(".amdhsa_kernel ")[1:]:
        name = blk.split("\n")[0].strip()
        dn = demangle(name)
        if os.environ["KRES_SEQ"] in dn:
            body = asm[asm.index(name + ":"):]
            print(dn)
            sequence(body[:body.index("s_endpgm")])
