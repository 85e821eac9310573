"""INTEGRATION.md promises that every code block of its sections 1-4 is a verbatim quote of COMPILED code -- the shipped
reference-side binding include/oalgpu_openal.hpp, or the bridge oracle/ref_bridge.cpp that compiles it against the reference.
This checks the promise: each ```cpp block must occur in one of the two files (indentation aside)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _norm(text):
    return "\n".join(line.strip() for line in text.strip().split("\n") if line.strip())


def test_integration_md_quotes_compiled_code_only():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sources = [_norm(open(os.path.join(ROOT, p)).read()) for p in ("include/oalgpu_openal.hpp", "oracle/ref_bridge.cpp")]
    blocks = re.findall(r"```cpp\n(.*?)```", doc, flags=re.S)
    assert len(blocks) >= 9
    for b in blocks:
        nb = _norm(b)
        assert any(nb in s for s in sources), "not a quote of compiled code:\n" + b[:400]
