"""TEST INFRASTRUCTURE -- compile the C oracle with gcc (no GPU, no hipcc involved).

    python -m oracle.build        ->  oracle/libt2h_oracle.so

The reference is pure Python (no C/C++ sources), so there is no ``oracle/_ref``
build: the "real reference" leg of the oracle is the Python import in
tests/golden/ref_import.py, whose outputs are the committed fixtures.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "t2h_oracle.c")
OUT = os.path.join(HERE, "libt2h_oracle.so")


def build(force: bool = False) -> str:
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-std=c11", "-fPIC", "-shared",
           "-fvisibility=hidden", "-Wall", "-Wextra", "-o", OUT, SRC, "-lm"]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
