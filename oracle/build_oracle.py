"""Builds the oracle's C restatement (gcc only; no GPU, no reference needed)."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libmas_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "mas_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", LIB, src, "-lm"])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
