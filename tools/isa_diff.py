#!/usr/bin/env python3
"""Which device kernels changed since a commit?  Cross-compiles every csrc/*.hip of <commit> and of the working tree
to gfx950 assembly (no GPU needed) and compares the instruction streams of the kernels, ignoring symbol names,
comments and label numbers:   python tools/isa_diff.py <commit>
Prints, per file, the kernels of the old tree whose code no longer exists verbatim in the new tree."""
import hashlib, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--cuda-device-only", "-S"]


def kernels(asm):
    out, name, body = {}, None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        if line.startswith(".Lfunc_end"):
            out[name] = hashlib.md5("\n".join(body).encode()).hexdigest()[:12]
            name = None
            continue
        t = re.sub(r";.*$", "", line).rstrip()
        t = re.sub(r"\.LBB\d+_", ".LBB_", t)
        if t.strip() and not t.lstrip().startswith((".amdhsa_kernel", ".section", ".p2align", ".type", ".globl", ".weak", ".protected")):
            body.append(t)
    return out


def compile_tree(src_root, tmp, tag):
    res = {}
    cs = os.path.join(src_root, "scalable-ccd_amd", "csrc")
    for f in sorted(os.listdir(cs)):
        if not f.endswith(".hip"):
            continue
        out = os.path.join(tmp, f"{tag}_{f}.s")
        subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-I" + os.path.join(src_root, "include"), "-I" + cs, "-o", out, os.path.join(cs, f)],
                       check=True, stderr=subprocess.DEVNULL)
        res[f] = kernels(open(out).read())
    return res


def main():
    commit = sys.argv[1]
    with tempfile.TemporaryDirectory() as tmp:
        old_root = os.path.join(tmp, "old")
        os.makedirs(old_root)
        tar = subprocess.run(["git", "-C", ROOT, "archive", commit, "scalable-ccd_amd/csrc", "include"], check=True, capture_output=True).stdout
        subprocess.run(["tar", "-x", "-C", old_root], input=tar, check=True)
        old, new = compile_tree(old_root, tmp, "old"), compile_tree(ROOT, tmp, "new")
    changed = 0
    for f, ko in old.items():
        have = set(new.get(f, {}).values())
        gone = [k for k, h in ko.items() if h not in have]
        print(f"{f}: {len(ko)} kernels at {commit}, {len(new.get(f, {}))} now, {len(gone)} changed or removed")
        for k in gone:
            print("   ", subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:110])
        changed += len(gone)
    print("kernels of", commit, "whose code is gone:", changed)


if __name__ == "__main__":
    main()
