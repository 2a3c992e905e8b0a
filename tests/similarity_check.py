#!/usr/bin/env python3
"""Token similarity of every product .py against the same-named file of the reference (comments and docstrings stripped,
difflib ratio) -- the check the round-3 review ran by hand.  Build-container tool: needs /root/reference (absent on the GPU
box); `tests/test_abi_and_host.py::test_host_files_are_not_transcriptions` runs it when the reference is there.

    python tests/similarity_check.py [--limit 0.6]
"""
import argparse
import difflib
import io
import os
import sys
import tokenize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mmwave_radar_processing_amd")
REF = "/root/reference/mmwave_radar_processing"


def tokens(path):
    out, prev = [], None
    with open(path, "rb") as fh:
        src = fh.read()
    try:
        for tok in tokenize.tokenize(io.BytesIO(src).readline):
            if tok.type in (tokenize.COMMENT, tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT, tokenize.ENCODING):
                continue
            if tok.type == tokenize.STRING and prev in (None, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT, ":"):
                prev = tokenize.STRING
                continue                      # a docstring (a string statement)
            out.append(tok.string)
            prev = tok.type if tok.string != ":" else ":"
    except tokenize.TokenError:
        pass
    return out


def scores():
    ref_by_name = {}
    for dirpath, _, files in os.walk(REF):
        if "archived" in dirpath:
            continue
        for f in files:
            if f.endswith(".py"):
                ref_by_name.setdefault(f, []).append(os.path.join(dirpath, f))
    rows = []
    for dirpath, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(".py") and f != "__init__.py" and f in ref_by_name:
                mine = os.path.join(dirpath, f)
                a = tokens(mine)
                for ref in ref_by_name[f]:
                    rows.append((difflib.SequenceMatcher(None, a, tokens(ref), autojunk=False).ratio(),
                                 os.path.relpath(mine, ROOT), os.path.relpath(ref, REF)))
    return sorted(rows, reverse=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--limit", type=float, default=0.6)
    args = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("reference not present")
    rows = scores()
    for r, mine, ref in rows:
        print(f"{r:.2f}  {mine}  vs  {ref}")
    sys.exit(1 if rows and rows[0][0] > args.limit else 0)
