"""SURVEY.md 8(c) "what travels": only the repo's own restatement, kernels and fixtures go to the GPU box -- nothing compiled
from /root/reference sources.  `gpurun` pushes the tree minus .git/, gpurun_out/ and the paths .gpurunignore lists, so every
output of a recipe that compiles reference SOURCES must sit under a path .gpurunignore (and .gitignore) names, and no binary
may be tracked by git."""
import fnmatch
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _patterns(name):
    with open(os.path.join(ROOT, name)) as f:
        return [ln.strip() for ln in f if ln.strip() and not ln.lstrip().startswith("#")]


def _ignored(rel, patterns):
    """gitignore-style match, as far as the two files use it: `dir/` prefixes, globs on the basename or the whole path."""
    for p in patterns:
        if p.endswith("/"):
            d = p.rstrip("/")
            if rel == d or rel.startswith(d + "/") or ("/" not in d and ("/" + d + "/") in ("/" + rel + "/")):
                return True
        elif fnmatch.fnmatch(rel, p) or fnmatch.fnmatch(os.path.basename(rel), p):
            return True
    return False


def test_reference_built_binaries_do_not_travel_and_are_not_tracked():
    push, track = _patterns(".gpurunignore"), _patterns(".gitignore")
    assert "oracle/_ref/" in push and "oracle/_ref/" in track
    # the one recipe that compiles reference sources writes only below oracle/_ref/
    with open(os.path.join(ROOT, "oracle", "Makefile")) as f:
        mk = f.read()
    outs = re.findall(r"\$\(REF\)/\S+\.cpp\s*\\?\s*-o (\S+)", mk.replace("\\\n", " "))
    assert outs and all(o.startswith("_ref/") for o in outs), outs
    for dirpath, _, files in os.walk(os.path.join(ROOT, "oracle", "_ref")):
        for fn in files:
            rel = os.path.relpath(os.path.join(dirpath, fn), ROOT)
            assert _ignored(rel, push), f"{rel} would be pushed to the GPU box"
            assert _ignored(rel, track), f"{rel} could be committed"
    # no other recipe in the tree compiles a reference SOURCE file (headers of the interface being implemented are
    # included by tests/cpp, which is the point of that harness)
    for dirpath, dirs, files in os.walk(ROOT):
        dirs[:] = [d for d in dirs if d not in (".git", "gpurun_out", "__pycache__", "_ref")]
        for fn in files:
            if fn == "Makefile" or fn.endswith((".sh", ".mk")):
                p = os.path.join(dirpath, fn)
                if os.path.relpath(p, ROOT) == os.path.join("oracle", "Makefile"):
                    continue
                with open(p, errors="replace") as f:
                    txt = f.read()
                assert not re.search(r"(/root/reference|\$\(REF\))/\S+\.(cpp|c|cu)\b", txt), p


def test_no_binary_is_tracked_by_git():
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        return
    files = subprocess.run(["git", "ls-files", "-z"], cwd=ROOT, capture_output=True, check=True).stdout.split(b"\0")
    bad = []
    for f in files:
        if not f:
            continue
        p = os.path.join(ROOT.encode(), f)
        if not os.path.isfile(p):
            continue
        with open(p, "rb") as fh:
            if fh.read(4) == b"\x7fELF":
                bad.append(f.decode())
    assert not bad, bad
