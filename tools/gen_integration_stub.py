#!/usr/bin/env python
"""Rewrite the generated part of INTEGRATION.md section 3 (the ctypes mirror of MomRasterArgs) from the binding in
iclr2025_3d-mom_amd/_native.py.  tests/test_abi.py fails when the two differ; run this after changing the struct."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BEGIN = "# --- generated from iclr2025_3d-mom_amd/_native.py"
END = "# --- end generated ---\n"


def generated_block(text):
    i = text.index(BEGIN)
    i = text.index("\n", i) + 1
    return i, text.index(END, i)


def main():
    N = importlib.import_module("iclr2025_3d-mom_amd._native")
    path = os.path.join(ROOT, "INTEGRATION.md")
    text = open(path).read()
    i, j = generated_block(text)
    new = text[:i] + N.ctypes_mirror_source(N.MomRasterArgs) + text[j:]
    if new != text:
        open(path, "w").write(new)
        print("INTEGRATION.md updated")
    else:
        print("INTEGRATION.md is current")


if __name__ == "__main__":
    main()
