#!/usr/bin/env python3
"""Generates the inline-asm blocks of the front end's packed-fp32 butterflies (csrc/fe_common.hpp: radix8_pk, pk_cmul7) — a list
of operations in issue order goes through a register allocator (a result may take the register of an operand that dies at that
instruction, else the register that has been free the longest) — and checks every block against the plain formulas with random fp32
values, operation by operation (each is one IEEE add / mul / fma per half, so the check is exact).
usage: python tools/gen/fe_blocks.py > webspeechanalyzer_amd/csrc/fe_blocks.inc"""
import numpy as np

F = np.float32


def fma(a, b, c):
    return F(np.float64(a) * np.float64(b) + np.float64(c))      # exact product of two fp32 in fp64; one rounding to fp32 (double rounding cannot occur: 48-bit product + fp32 addend fits 53 bits unless exponents are far apart — the check only needs self-consistency: both sides use this function)


# semantic of every packed op on (x, y) pairs; mirrors fe_common.hpp
SEM = {
    "ADD": lambda a, b: (F(a[0] + b[0]), F(a[1] + b[1])),
    "SUB": lambda a, b: (F(a[0] - b[0]), F(a[1] - b[1])),
    "ADDMI": lambda a, b: (F(a[0] + b[1]), F(a[1] - b[0])),
    "SUBMI": lambda a, b: (F(a[0] - b[1]), F(a[1] + b[0])),
    "W8A": lambda t: (F(t[0] + t[1]), F(t[1] - t[0])),
    "W83A": lambda t: (F(t[1] - t[0]), F(-t[0] - t[1])) if False else (F(t[1] - t[0]), F(t[0] + t[1])),
    "W8M": lambda u, ss: (F(u[0] * ss[0]), F(u[1] * ss[1])),
    "W83M": lambda u, ss: (F(u[0] * ss[0]), F(-(u[1] * ss[1]))),
    "CM": lambda x, w: (F(x[0] * w[0]), F(x[0] * w[1])),
    "CF": lambda x, w, t: (fma(F(-x[1]), w[1], t[0]), fma(x[1], w[0], t[1])),
    "ADDC": lambda a, b: (F(a[0] + b[0]), F(a[1] - b[1])),
    "SUBC": lambda a, b: (F(a[0] - b[0]), F(a[1] + b[1])),
    "SQ": lambda x: (F(x[0] * x[0]), F(x[1] * x[1])),
    "PW": lambda x, q: (fma(x[0], x[0], q[1]), fma(x[0], x[0], q[1])),
}
TXT = {
    "ADD": "v_pk_add_f32 {d}, {a}, {b}",
    "SUB": "v_pk_add_f32 {d}, {a}, {b} neg_lo:[0,1] neg_hi:[0,1]",
    "ADDMI": "v_pk_add_f32 {d}, {a}, {b} op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]",
    "SUBMI": "v_pk_add_f32 {d}, {a}, {b} op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]",
    "W8A": "v_pk_add_f32 {d}, {a}, {a} op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]",
    "W83A": "v_pk_add_f32 {d}, {a}, {a} op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[0,1]",
    "W8M": "v_pk_mul_f32 {d}, {a}, {b}",
    "W83M": "v_pk_mul_f32 {d}, {a}, {b} neg_hi:[0,1]",
    "CM": "v_pk_mul_f32 {d}, {a}, {b} op_sel:[0,0] op_sel_hi:[0,1]",
    "CF": "v_pk_fma_f32 {d}, {a}, {b}, {c} op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]",
    "ADDC": "v_pk_add_f32 {d}, {a}, {b} neg_hi:[0,1]",
    "SUBC": "v_pk_add_f32 {d}, {a}, {b} neg_lo:[0,1]",
    "SQ": "v_pk_mul_f32 {d}, {a}, {a}",
    "PW": "v_pk_fma_f32 {d}, {a}, {a}, {b} op_sel:[0,0,1] op_sel_hi:[0,0,1]",
}


def allocate(ops, inputs, consts, outputs, n_regs):
    """ops: [(dst value, op, src values...)]; inputs: values that arrive in registers r0..; consts: read-only operands (own
    registers, never reused); n_regs: rotating registers in all.  Returns (asm lines, {output value: register})."""
    last_use = {}
    for k, (d, op, *src) in enumerate(ops):
        for s_ in src:
            last_use[s_] = k
    for o in outputs:
        last_use[o] = len(ops)
    where = {v: i for i, v in enumerate(inputs)}
    free = [i for i in range(len(inputs), n_regs)]
    lines = []
    for k, (d, op, *src) in enumerate(ops):
        dying = [where[s_] for s_ in src if s_ not in consts and last_use[s_] == k]
        names = {}
        for key, s_ in zip("abc", src):
            names[key] = "%%[%s]" % s_ if s_ in consts else "%%[r%d]" % where[s_]
        if free:
            reg = free.pop(0)                     # the register that has been free the longest: producers stay far from earlier readers
        elif dying:
            reg = dying.pop(0)
        else:
            raise RuntimeError("out of registers at op %d" % k)
        for r in dying:
            free.append(r)
        for s_ in src:
            if s_ not in consts and last_use[s_] == k:
                del where[s_]
        where[d] = reg
        lines.append(TXT[op].format(d="%%[r%d]" % reg, **names))
    return lines, {o: where[o] for o in outputs}


def simulate(lines_ops, ops, inputs, consts, outputs, n_regs, alloc_lines, out_map, rng):
    vals = {v: (F(rng.standard_normal()), F(rng.standard_normal())) for v in list(inputs) + list(consts)}
    ref = dict(vals)
    for d, op, *src in ops:
        ref[d] = SEM[op](*[ref[s_] for s_ in src])
    # replay on registers by parsing the allocation again
    regs = {i: vals[v] for i, v in enumerate(inputs)}
    last_use, where = {}, {v: i for i, v in enumerate(inputs)}
    import re
    for line, (d, op, *src) in zip(alloc_lines, ops):
        toks = re.findall(r"%\[(\w+)\]", line)
        dst, srcs = toks[0], toks[1:]
        if op in ("W8A", "W83A", "SQ"):
            srcs = srcs[:1]
        if op == "PW":
            srcs = [srcs[0], srcs[2]]
        get = lambda t: regs[int(t[1:])] if t.startswith("r") and t[1:].isdigit() else vals[t]
        regs[int(dst[1:])] = SEM[op](*[get(t) for t in srcs])
    for o, r in out_map.items():
        a, b = regs[r], ref[o]
        assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes(), (o, a, b)


R8_FULL = [
    ("a0", "ADD", "v0", "v4"), ("b0", "SUB", "v0", "v4"), ("a1", "ADD", "v1", "v5"), ("x1", "SUB", "v1", "v5"),
    ("a3", "ADD", "v3", "v7"), ("x3", "SUB", "v3", "v7"), ("a2", "ADD", "v2", "v6"), ("t2", "SUB", "v2", "v6"),
    ("y1", "W8A", "x1"), ("y3", "W83A", "x3"), ("c0", "ADD", "a0", "a2"), ("c2", "SUB", "a0", "a2"),
    ("b1", "W8M", "y1", "ss"), ("b3", "W83M", "y3", "ss"), ("c1", "ADD", "a1", "a3"), ("u3", "SUB", "a1", "a3"),
    ("d0", "ADDMI", "b0", "t2"), ("d2", "SUBMI", "b0", "t2"), ("d1", "ADD", "b1", "b3"), ("w3", "SUB", "b1", "b3"),
    ("o0", "ADD", "c0", "c1"), ("o4", "SUB", "c0", "c1"), ("o2", "ADDMI", "c2", "u3"), ("o6", "SUBMI", "c2", "u3"),
    ("o1", "ADD", "d0", "d1"), ("o5", "SUB", "d0", "d1"), ("o3", "ADDMI", "d2", "w3"), ("o7", "SUBMI", "d2", "w3"),
]
R8_HALF = [          # v4 .. v7 are structural zeros: a_k = b_k' = v_k
    ("y1", "W8A", "v1"), ("y3", "W83A", "v3"), ("c0", "ADD", "v0", "v2"), ("c2", "SUB", "v0", "v2"),
    ("c1", "ADD", "v1", "v3"), ("u3", "SUB", "v1", "v3"), ("d0", "ADDMI", "v0", "v2"), ("d2", "SUBMI", "v0", "v2"),
    ("b1", "W8M", "y1", "ss"), ("b3", "W83M", "y3", "ss"), ("o0", "ADD", "c0", "c1"), ("o4", "SUB", "c0", "c1"),
    ("d1", "ADD", "b1", "b3"), ("w3", "SUB", "b1", "b3"), ("o2", "ADDMI", "c2", "u3"), ("o6", "SUBMI", "c2", "u3"),
    ("o1", "ADD", "d0", "d1"), ("o5", "SUB", "d0", "d1"), ("o3", "ADDMI", "d2", "w3"), ("o7", "SUBMI", "d2", "w3"),
]
CMUL7 = [("m1", "CM", "x1", "w1"), ("m2", "CM", "x2", "w2"), ("m3", "CM", "x3", "w3"), ("m4", "CM", "x4", "w4"),
         ("y1", "CF", "x1", "w1", "m1"), ("m5", "CM", "x5", "w5"), ("y2", "CF", "x2", "w2", "m2"), ("m6", "CM", "x6", "w6"),
         ("y3", "CF", "x3", "w3", "m3"), ("m7", "CM", "x7", "w7"), ("y4", "CF", "x4", "w4", "m4"), ("y5", "CF", "x5", "w5", "m5"),
         ("y6", "CF", "x6", "w6", "m6"), ("y7", "CF", "x7", "w7", "m7")]


SPLIT5 = ([x for c in range(5) for x in (("e%d" % c, "ADDC", "a%d" % c, "b%d" % c), ("o%d" % c, "SUBC", "a%d" % c, "b%d" % c))]
          + [("t0", "CM", "o0", "w0"), ("t1", "CM", "o1", "w1"), ("t2", "CM", "o2", "w2"), ("u0", "CF", "o0", "w0", "t0"), ("t3", "CM", "o3", "w3"),
             ("u1", "CF", "o1", "w1", "t1"), ("t4", "CM", "o4", "w4"), ("u2", "CF", "o2", "w2", "t2"), ("u3", "CF", "o3", "w3", "t3"), ("u4", "CF", "o4", "w4", "t4")]
          + [("x%d" % c, "ADDMI", "e%d" % c, "u%d" % c) for c in range(5)] + [("q%d" % c, "SQ", "x%d" % c) for c in range(5)]
          + [("p%d" % c, "PW", "x%d" % c, "q%d" % c) for c in range(5)])


def emit(fn, comment, ops, inputs, consts, outputs, n_regs, in_expr, const_expr, out_stmt):
    """in_expr(value) / const_expr(value): the C++ expression an input / a read-only operand comes from; out_stmt(value, reg var): the
    C++ statement that stores a result."""
    lines, out_map = allocate(ops, inputs, consts, outputs, n_regs)
    rng = np.random.default_rng(1)
    for _ in range(50):
        simulate(None, ops, inputs, consts, outputs, n_regs, lines, out_map, rng)
    print("// %s: %d instructions, %d rotating registers" % (comment, len(lines), n_regs))
    print("#define %s \\" % fn)
    print("    do { \\")
    print("        v2f %s; \\" % ", ".join(("r%d = %s" % (i, in_expr(inputs[i]))) if i < len(inputs) else "r%d" % i for i in range(n_regs)))
    print("        asm( \\")
    for ln in lines:
        print('            "%s\\n\\t" \\' % ln.replace("%%", "%"))
    outs = ", ".join('[r%d] "%s"(r%d)' % (i, "+v" if i < len(inputs) else "=&v", i) for i in range(n_regs))
    ins = ", ".join('[%s] "v"(%s)' % (c, const_expr(c)) for c in sorted(consts))
    print("            : %s \\" % outs)
    print("            : %s); \\" % ins)
    print("        %s \\" % " ".join(out_stmt(o, "r%d" % out_map[o]) for o in outputs))
    print("    } while (0)")
    print()


if __name__ == "__main__":
    print("// fe_blocks.inc — GENERATED by tools/gen/fe_blocks.py (operation lists, register allocation and the self-check live there); do not edit.")
    print("// The packed-fp32 butterflies of the front end as single asm blocks (why: fe_common.hpp).  v, w, ss, za, zb, pw are the caller's names.")
    print()
    idx = lambda v: int(v[1:])
    emit("WSA_R8_FULL(v, ss)", "radix-8 DIF butterfly, eight inputs (radix8_pk<8>)", R8_FULL, ["v%d" % i for i in range(8)], {"ss"}, ["o%d" % i for i in range(8)], 10,
         lambda v: "(v)[%d]" % idx(v), lambda c: "(ss)", lambda o, r: "(v)[%d] = %s;" % (idx(o), r))
    emit("WSA_R8_HALF(v, ss)", "radix-8 DIF butterfly, inputs 4 .. 7 structural zeros (radix8_pk<NZ <= 4>)", R8_HALF, ["v%d" % i for i in range(4)], {"ss"}, ["o%d" % i for i in range(8)], 10,
         lambda v: "(v)[%d]" % idx(v), lambda c: "(ss)", lambda o, r: "(v)[%d] = %s;" % (idx(o), r))
    emit("WSA_CMUL7(v, w)", "v[k] *= w[k], k = 1 .. 7", CMUL7, ["x%d" % i for i in range(1, 8)], {"w%d" % i for i in range(1, 8)}, ["y%d" % i for i in range(1, 8)], 11,
         lambda v: "(v)[%d]" % idx(v), lambda c: "(w)[%d]" % idx(c), lambda o, r: "(v)[%d] = %s;" % (idx(o), r))
    emit("WSA_SPLIT5(za, zb, w, pw)", "real-FFT split + power of five rows: e = za + conj(zb), o = za - conj(zb), t = o w, x = e + (-i) t, pw = fma(x.x, x.x, x.y x.y)", SPLIT5,
         ["a%d" % c for c in range(5)] + ["b%d" % c for c in range(5)], {"w%d" % c for c in range(5)}, ["p%d" % c for c in range(5)], 13,
         lambda v: "(%s)[%d]" % ("za" if v[0] == "a" else "zb", idx(v)), lambda c: "(w)[%d]" % idx(c), lambda o, r: "(pw)[%d] = %s.x;" % (idx(o), r))
