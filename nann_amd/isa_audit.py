"""Scan AMDGPU assembly (hipcc --save-temps .s) for VGPR spill code that the register
allocator placed in a basic block BEFORE the instruction that restores the exec mask
(`s_or_b64 exec, exec, ...`): such a spill only stores the lanes that are still enabled and
the later reload returns garbage for the others.  Seen with ROCm 7.2 hipcc on kernels with
heavy spilling around divergent loops.  Blocks the compiler names "Flow" (the join blocks of divergent regions, where the restore
must come first) are the confirmed miscompile; other hits may be blocks nested inside a
masked region, which is legal.  build.py runs this on every object and refuses to link on a
Flow hit.  CLI: python -m nann_amd.isa_audit file.s [...]"""
import re
import sys


NARROW = re.compile(r"s_(and|andn2|xor|or)_saveexec_b64|s_(and|andn2|xor)_b64\s+exec,|s_cmov_b64\s+exec,")
RESTORE = re.compile(r"s_or_b64\s+exec,\s*exec,|s_mov_b64\s+exec,")


def audit(path):
    """Hits: VGPR spill stores / reloads that sit between a block label and the exec
    restore that opens the block (no exec-narrowing instruction in between)."""
    hits = []
    kernel, block = None, None
    pending, open_block = [], False  # open_block: no exec write seen yet in this block
    for no, line in enumerate(open(path, errors="replace"), 1):
        s = line.strip()
        m = re.match(r"^([A-Za-z_.$][\w.$]*):", line)
        if m:
            name = m.group(1)
            if not name.startswith(".L"):
                kernel = name
            block, pending, open_block = name + (" %Flow" if "%Flow" in line else ""), [], True
            continue
        if not s or s.startswith(";") or s.startswith("."):
            continue
        if not open_block:
            continue
        if RESTORE.match(s):
            hits.extend((kernel, block, n, t) for n, t in pending)
            open_block = False
        elif NARROW.search(s) or s.startswith("s_cbranch") or s.startswith("s_branch"):
            open_block = False
        elif ("scratch_store" in s or "scratch_load" in s) and ("Spill" in s or "Reload" in s):
            pending.append((no, s))
    return hits


def flow_hits(path):
    return [h for h in audit(path) if "%Flow" in h[1]]


if __name__ == "__main__":
    bad = 0
    for p in sys.argv[1:]:
        for kernel, block, no, text in audit(p):
            print(f"{p}:{no}: [{kernel} {block}] {text}")
            bad += 1
    print(f"{bad} spill instruction(s) ahead of an exec restore")
    sys.exit(1 if bad else 0)
