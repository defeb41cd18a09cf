#!/usr/bin/env python3
"""One-line-per-basic-block summary of a kernel's instruction schedule in a hipcc assembly dump: where the MFMAs, the memory
instructions and the waits ended up after LLVM's passes (how the machine-sink finding of build.py's TU_FLAGS was made).

    cd parrot_tts_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only tu_split16.hip -o /tmp/tu16.s
    python tools/isa_schedule.py /tmp/tu16.s Li2ELi2ELi4ELi4ELi11ELi2E      # (a substring of the mangled kernel name)

M MFMA · L buffer_load_dwordx4 (weights) · l other buffer loads · S buffer stores · r / w ds_read / ds_write ·
[vm(n) lgk(n)] s_waitcnt · |B| s_barrier · <…> branches · !SCRATCH! spill traffic.  Blocks without MFMAs or barriers are skipped."""
import re
import sys


def main():
    text, name = open(sys.argv[1]).read(), sys.argv[2]
    m = re.search(r"^(_ZN6parrot\w*" + re.escape(name) + r"\w*):.*?\n(.*?)s_endpgm", text, re.S | re.M)
    if not m:
        sys.exit(f"no kernel matching {name!r}")
    lines = m.group(2).split("\n")
    print(m.group(1), f"({len(lines)} lines)")
    seq = []
    for ln in lines:
        s = ln.strip()
        op = s.split(" ")[0].split("\t")[0]
        if re.match(r"^\.LBB\d+_\d+:", s):
            seq.append("\n" + s.split(";")[0] + " ")
        elif op.startswith(("s_cbranch", "s_branch")):
            seq.append(" <" + s.replace("\t", " ") + "> ")
        elif op.startswith("v_mfma"):
            seq.append("M")
        elif op.startswith("buffer_load"):
            seq.append("D" if " lds" in s else ("L" if "dwordx4" in s else "l"))
        elif op.startswith("buffer_store"):
            seq.append("S")
        elif op.startswith("ds_read"):
            seq.append("r")
        elif op.startswith("ds_write"):
            seq.append("w")
        elif op.startswith("s_waitcnt"):
            seq.append("[" + s.replace("s_waitcnt ", "").replace("vmcnt", "vm").replace("lgkmcnt", "lgk") + "]")
        elif op.startswith("s_barrier"):
            seq.append("|B|")
        elif op.startswith("scratch_"):
            seq.append("!SCRATCH!")
    for blk in "".join(seq).split("\n"):
        if "M" in blk or "|B|" in blk:
            print(blk)


if __name__ == "__main__":
    main()
