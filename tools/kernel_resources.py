"""Registers, scratch, LDS and occupancy of every kernel of the library, from the compiler's own remarks
(-Rpass-analysis=kernel-resource-usage), one line per kernel; exit code 1 if a hand-written kernel (anything but rocPRIM's)
spills or uses scratch memory -- k_hamm64_mfma<true> carried two spilled lane constants through half of round 6, reloaded in
front of every candidate list (NOTES 13.3b).  No GPU needed.
    python tools/kernel_resources.py [file.hip ...]"""
import glob, os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cbird_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I../../include", "-mllvm",
         "-amdgpu-mfma-vgpr-form=1", "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null"]
PAT = re.compile(r"Function Name: (\S+).*?TotalSGPRs: (\d+).*?VGPRs: (\d+).*?AGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?"
                 r"Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+).*?LDS Size \[bytes/block\]: (\d+)", re.S)


def one(f):
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [f], cwd=ROOT, capture_output=True, text=True)
    return f, r.stderr


def main():
    files = sys.argv[1:] or sorted(os.path.basename(p) for p in glob.glob(os.path.join(ROOT, "*.hip")))
    bad = 0
    with ThreadPoolExecutor(4) as ex:
        for f, txt in ex.map(one, files):
            for m in PAT.finditer(txt):
                name, sg, vg, ag, sc, occ, ssp, vsp, lds = m.groups()
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
                short = dem.replace("cbh::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")
                mm = re.match(r"([\w:]+(?:<[^(]*?>)?)\(", short)
                short = (mm.group(1) if mm else short)[:70]
                flag = ""
                if (int(sc) or int(vsp) or int(ssp)) and "rocprim" not in dem:
                    flag, bad = "  <-- scratch / spill", bad + 1
                print(f"{f:22s} {short:70s} VGPR {vg:>3s} AGPR {ag:>3s} SGPR {sg:>3s} scratch {sc:>3s} LDS {lds:>6s} waves/SIMD {occ}{flag}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
