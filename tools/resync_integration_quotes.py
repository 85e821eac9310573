"""INTEGRATION.md's ```cpp blocks are verbatim quotes of include/oalgpu_openal.hpp / oracle/ref_bridge.cpp (tests/test_docs_quotes.py).
After an edit of those files: re-quote every block that no longer matches -- the stretch of the source between the block's first and
last line (which must still exist) replaces it.  Prints what it changed; review the diff."""
import os, re, sys, textwrap
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
doc_path = os.path.join(ROOT, "INTEGRATION.md")
doc = open(doc_path).read()
srcs = {p: open(os.path.join(ROOT, p)).read().split("\n") for p in ("include/oalgpu_openal.hpp", "oracle/ref_bridge.cpp")}
norm = lambda t: "\n".join(l.strip() for l in t.strip().split("\n") if l.strip())
changed = 0
def fix(m):
    global changed
    block = m.group(1)
    nb = norm(block)
    if any(nb in norm("\n".join(lines)) for lines in srcs.values()):
        return m.group(0)
    bl = [l.strip() for l in block.split("\n") if l.strip()]
    first, last = bl[0], bl[-1]
    for p, lines in srcs.items():
        st = [l.strip() for l in lines]
        for i, l in enumerate(st):
            if l != first: continue
            # the end: the occurrence of the block's last line whose stretch resembles the old quote most
            import difflib
            best = None
            for j in range(i + 1, min(len(st), i + 2 * len(bl) + 60)):
                if st[j] == last:
                    cand = [x for x in st[i:j + 1] if x]
                    r = difflib.SequenceMatcher(None, bl, cand, autojunk=False).ratio()
                    if best is None or r > best[0]: best = (r, j)
            if best:
                j = best[1]
                new = textwrap.dedent("\n".join(lines[i:j + 1]))
                changed += 1
                print(f"re-quoted {p}:{i + 1}-{j + 1} ({len(bl)} -> {len([x for x in st[i:j + 1] if x])} lines, similarity {best[0]:.2f}), block starting: {first[:80]}")
                return "```cpp\n" + new + "\n```"
    print("NOT FOUND:", first[:100]); return m.group(0)
doc2 = re.sub(r"```cpp\n(.*?)```", fix, doc, flags=re.S)
if changed: open(doc_path, "w").write(doc2)
print(changed, "blocks re-quoted")
