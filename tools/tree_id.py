#!/usr/bin/env python3
"""Identity of the source tree an evidence file was taken from.  The GPU box has no .git (gpurun ships a snapshot), so the id is a
SHA-1 over the tracked-looking sources (package, csrc, bench.py, tools/, tests/, include/, oracle/) in sorted path order; tools/tree_id.py
run in the repo prints the same id for the same sources, so a profiles/ file can be matched to a commit (`git stash; python
tools/tree_id.py` at that commit).  --stamp <id> files...: records the id inside each file (JSON: key "tree_id"; CSV / text: a
trailing comment line)."""
import hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXT = (".py", ".hip", ".h", ".sh", ".md", "Makefile")
DIRS = ("neural_marionette_amd", "tools", "tests", "include", "oracle")


def tree_id():
    files = [os.path.join(ROOT, f) for f in ("bench.py", "__graft_entry__.py")]
    for d in DIRS:
        for base, dirs, names in os.walk(os.path.join(ROOT, d)):
            dirs[:] = sorted(x for x in dirs if x not in ("__pycache__", "golden", "_ref"))
            for n in sorted(names):
                if n.endswith(EXT) and not n.startswith("_run"):
                    files.append(os.path.join(base, n))
    h = hashlib.sha1()
    for f in sorted(files):
        h.update(os.path.relpath(f, ROOT).encode()); h.update(b"\0")
        h.update(open(f, "rb").read()); h.update(b"\0")
    return h.hexdigest()[:16]


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--stamp":
        tid = sys.argv[2]
        for f in sys.argv[3:]:
            if not os.path.isfile(f):
                continue
            if f.endswith(".json"):
                try:
                    d = json.load(open(f))
                    if isinstance(d, dict):
                        d["tree_id"] = tid
                        json.dump(d, open(f, "w"), indent=1)
                        continue
                except Exception:
                    pass
            if f.endswith(".json.log"):
                try:
                    d = json.loads(open(f).read().strip().splitlines()[-1]); d["tree_id"] = tid
                    open(f, "w").write(json.dumps(d) + "\n")
                    continue
                except Exception:
                    pass
            open(f, "a").write("# tree_id %s (tools/tree_id.py)\n" % tid)
    else:
        print(tree_id())


if __name__ == "__main__":
    main()
