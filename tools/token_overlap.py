#!/usr/bin/env python3
"""token_overlap.py MINE REF [REF ...] — share of MINE's tokens that lie in runs of >= 6 tokens also found in a REF file
(comments stripped; identifiers, numbers, strings and punctuation are tokens).  A self-check against renamed copies."""
import re
import sys

TOK = re.compile(r"[A-Za-z_$][A-Za-z0-9_$]*|\d+\.?\d*|'[^']*'|\"[^\"]*\"|`[^`]*`|[^\sA-Za-z0-9_$]")


def tokens(path):
    s = open(path, errors="replace").read()
    s = re.sub(r"/\*.*?\*/", " ", s, flags=re.S)
    s = re.sub(r"(?m)//.*$", " ", s)
    return TOK.findall(s)


def main():
    mine = tokens(sys.argv[1])
    n = 6
    grams = set()
    for ref in sys.argv[2:]:
        t = tokens(ref)
        for i in range(len(t) - n + 1):
            grams.add(tuple(t[i:i + n]))
    hit = [False] * len(mine)
    for i in range(len(mine) - n + 1):
        if tuple(mine[i:i + n]) in grams:
            for j in range(i, i + n):
                hit[j] = True
    print(f"{sys.argv[1]}: {sum(hit)} of {len(mine)} tokens in shared runs >= {n} = {100.0 * sum(hit) / max(1, len(mine)):.1f} %")


if __name__ == "__main__":
    main()
