"""Every file under profiles/ is named by profiles/README.md -- literally, or by one of the README's patterns (`*`, `{a,b}`
alternatives, `N` for a digit): a summary nobody can find the origin of is not evidence."""
import itertools
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROFILES = os.path.join(ROOT, "profiles")


def _expand(pattern):
    """`a{b,c}d` -> abd, acd (nested braces are not used)"""
    parts = re.split(r"(\{[^{}]*\})", pattern)
    choices = [p[1:-1].split(",") if p.startswith("{") else [p] for p in parts]
    return ["".join(c) for c in itertools.product(*choices)]


def _regex(pattern):
    out = ""
    for ch in pattern:
        if ch == "*":
            out += ".*"
        elif ch == "N":
            out += "[0-9]"
        else:
            out += re.escape(ch)
    return re.compile(out + r"\Z")


def listed_patterns():
    text = open(os.path.join(PROFILES, "README.md")).read()
    pats = []
    for line in text.split("\n"):
        if not line.startswith("| `"):
            continue
        first = line.split("|")[1]
        for tok in re.findall(r"`([^`]+)`", first):
            pats.extend(_expand(tok.strip()))
    return [_regex(p) for p in pats]


def test_every_profile_file_is_listed_in_the_readme():
    pats = listed_patterns()
    assert len(pats) > 60
    missing = []
    for base, _, files in os.walk(PROFILES):
        for f in files:
            rel = os.path.relpath(os.path.join(base, f), PROFILES)
            if rel == "README.md":
                continue
            if not any(p.match(rel) for p in pats):
                missing.append(rel)
    assert not missing, sorted(missing)
