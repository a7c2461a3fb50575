"""ORACLE (test infrastructure): the slice of numeric.js 1.2.6 that formantanalyzer's make_coeffs uses (ref
/root/reference/dist/main.js:2, inner module 5 @B38281: dotVV @B48151, dotMMsmall @B47148, inv @B55496, transpose
@B56450, tensor @B58719, norm2 / mapreduce @B57439, gradient @B89174, uncmin @B89779), restated loop for loop so that
every sum runs in the same order.  Python floats are IEEE doubles, like JavaScript Numbers."""
import math

EPS = 2220446049250313e-31           # numeric.epsilon


def _div(a, b):
    """JavaScript `/`: IEEE division, no exception (a singular normal matrix gives Infinity / NaN, which numeric.uncmin
    later turns into `throw new Error("uncmin: f(x0) is a NaN!")`)."""
    if b == 0:
        if a != a or a == 0:
            return float("nan")
        neg = (a < 0) != (math.copysign(1.0, b) < 0)
        return float("-inf") if neg else float("inf")
    return a / b


def dotVV(x, y):
    n = len(x)
    r = x[n - 1] * y[n - 1]
    i = n - 2
    while i >= 1:
        r += x[i] * y[i] + x[i - 1] * y[i - 1]
        i -= 2
    if i == 0:
        r += x[0] * y[0]
    return r


def dotMM(a, b):                     # dotMMsmall and dotMMbig add in the same order
    cols = [[b[k][j] for k in range(len(b))] for j in range(len(b[0]))]
    return [[dotVV(row, col) for col in cols] for row in a]


def dotMV(a, x):
    return [dotVV(row, x) for row in a]


def dotVM(x, b):
    return [dotVV(x, [b[k][j] for k in range(len(b))]) for j in range(len(b[0]))]


def transpose(a):
    return [[a[i][j] for i in range(len(a))] for j in range(len(a[0]))]


def identity(n):
    return [[1.0 if i == j else 0.0 for j in range(n)] for i in range(n)]


def inv(a):
    """Gauss-Jordan with partial pivoting exactly as numeric.inv (row operations on A and I in its loop order)."""
    m, n = len(a), len(a[0])
    A = [list(map(float, r)) for r in a]
    I = identity(m)
    for j in range(n):
        i0, v0 = -1, -1.0
        for i in range(j, m):
            k = abs(A[i][j])
            if k > v0:
                i0, v0 = i, k
        if i0 < 0:                               # all candidates NaN: `d[-1]` is undefined in JS -> TypeError
            raise ValueError("inv: no pivot")
        Aj = A[i0]; A[i0] = A[j]; A[j] = Aj
        Ij = I[i0]; I[i0] = I[j]; I[j] = Ij
        x = Aj[j]
        for k in range(j, n):
            Aj[k] = _div(Aj[k], x)
        for k in range(n - 1, -1, -1):
            Ij[k] = _div(Ij[k], x)
        for i in range(m - 1, -1, -1):
            if i != j:
                Ai, Ii = A[i], I[i]
                x = Ai[j]
                for k in range(j + 1, n):
                    Ai[k] -= Aj[k] * x
                for k in range(n - 1, -1, -1):
                    Ii[k] -= Ij[k] * x
    return I


def tensor(x, y):
    return [[xi * yj for yj in y] for xi in x]


def norm2(x):
    acc = 0.0
    for i in range(len(x) - 1, -1, -1):
        acc += x[i] * x[i]
    return math.sqrt(acc)


def gradient(f, x):
    n = len(x)
    f0 = f(x)
    if f0 != f0:
        raise ValueError("gradient: f(x) is a NaN!")
    x0 = list(x)
    J = [0.0] * n
    it = 0
    for i in range(n):
        h = max(1e-6 * f0, 1e-8)
        while True:
            it += 1
            if it > 20:
                raise ValueError("Numerical gradient fails")
            x0[i] = x[i] + h
            f1 = f(x0)
            x0[i] = x[i] - h
            f2 = f(x0)
            x0[i] = x[i]
            if f1 != f1 or f2 != f2:
                h /= 16
                continue
            J[i] = _div(f1 - f2, 2 * h)
            t0, t1, t2 = x[i] - h, x[i], x[i] + h
            d1 = _div(f1 - f0, h)
            d2 = _div(f0 - f2, h)
            N = max(abs(J[i]), abs(f0), abs(f1), abs(f2), abs(t0), abs(t1), abs(t2), 1e-8)
            errest = min(_div(max(abs(d1 - J[i]), abs(d2 - J[i]), abs(d1 - d2)), N), _div(h, N))
            if errest > 1e-3:
                h /= 16
            else:
                break
    return J


def uncmin(f, x0, tol=1e-8, maxit=1000):
    """BFGS with backtracking as numeric.uncmin; returns the solution vector."""
    tol = max(tol, EPS)
    x0 = list(x0)
    n = len(x0)
    f0 = f(x0)
    if f0 != f0:
        raise ValueError("uncmin: f(x0) is a NaN!")
    H1 = identity(n)
    it = 0
    g0 = gradient(f, x0)
    finite = lambda v: all(math.isfinite(t) for t in v)
    while it < maxit:
        if not finite(g0):
            break
        step = [-t for t in dotMV(H1, g0)]
        if not finite(step):
            break
        nstep = norm2(step)
        if nstep < tol:
            break
        t = 1.0
        df0 = dotVV(g0, step)
        x1 = x0
        s = None
        f1 = None
        while it < maxit:
            if t * nstep < tol:
                break
            s = [p * t for p in step]
            x1 = [a + b for a, b in zip(x0, s)]
            f1 = f(x1)
            if f1 - f0 >= 0.1 * t * df0 or f1 != f1:
                t *= 0.5
                it += 1
                continue
            break
        if t * nstep < tol:
            break
        if it == maxit:
            break
        g1 = gradient(f, x1)
        y = [a - b for a, b in zip(g1, g0)]
        ys = dotVV(y, s)
        Hy = dotMV(H1, y)
        c = _div(ys + dotVV(y, Hy), ys * ys)
        A = tensor(s, s)
        B1, B2 = tensor(Hy, s), tensor(s, Hy)
        H1 = [[(H1[i][j] + c * A[i][j]) - _div(B1[i][j] + B2[i][j], ys) for j in range(n)] for i in range(n)]
        x0, f0, g0 = x1, f1, g1
        it += 1
    return x0
