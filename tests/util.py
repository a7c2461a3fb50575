"""Shared helpers for the parity tests."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def jsnum(x):
    """ref_driver.js writes non-finite Numbers as strings ("NaN", "Infinity", "-Infinity")."""
    return float(x) if isinstance(x, str) else x


def jsvec(v):
    return np.array([jsnum(x) for x in v], dtype=np.float64)


def same_f64(a, b):
    """bitwise-equal doubles, NaN == NaN."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


def rel_err(a, b, abs_floor=1e-6):
    """max relative error with an absolute floor; NaN/Inf positions must coincide."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return np.inf
    fin = np.isfinite(a) & np.isfinite(b)
    if not bool(((a == b) | (np.isnan(a) & np.isnan(b)))[~fin].all()):
        return np.inf
    if not fin.any():
        return 0.0
    return float(np.max(np.abs(a[fin] - b[fin]) / np.maximum(np.abs(b[fin]), abs_floor)))


def load_backend_golden():
    spectra = np.load(os.path.join(GOLDEN, "backend_spectra.npz"))
    cases = json.load(open(os.path.join(GOLDEN, "backend_expected.json")))["cases"]
    return spectra, cases


def callbacks_equal(level, ref_cbs, got_cbs, exact=True, tol=1e-4):
    """Compare the callback sequences (reference JSON form vs ours).  Indices / timestamps always
    exact; feature vectors bit-exact (exact=True) or within tol relative error."""
    if len(ref_cbs) != len(got_cbs):
        return False, f"callback count {len(ref_cbs)} != {len(got_cbs)}"
    for r, o in zip(ref_cbs, got_cbs):
        if r[0] != o[0]:
            return False, f"si {r[0]} != {o[0]}"
        if level == 3:                 # (si, label, tracks): integers and exact doubles only (velocities are k/2, k/3)
            if json.loads(json.dumps(r[2])) != json.loads(json.dumps(o[2])):
                return False, f"si {r[0]} ranked tracks differ"
        elif level == 12:
            if [list(x) for x in r[2]] != [list(x) for x in o[2]]:
                return False, f"si {r[0]} syllable times {r[2]} != {o[2]}"
            if len(r[3]) != len(o[3]):
                return False, f"si {r[0]} row count {len(r[3])} != {len(o[3])}"
            for a, b in zip(r[3], o[3]):
                ok = same_f64(jsvec(a), b) if exact else rel_err(b, jsvec(a)) <= tol
                if not ok:
                    return False, f"si {r[0]} polynomial rows differ"
        elif level == 5:
            if not same_f64(jsvec(r[2]), o[2]):
                return False, f"si {r[0]} time {r[2]} != {list(o[2])}"
            ok = same_f64(jsvec(r[3]), o[3]) if exact else rel_err(o[3], jsvec(r[3])) <= tol
            if not ok:
                return False, f"si {r[0]} features differ"
        elif level == 13:
            if [list(x) for x in r[2]] != [list(x) for x in o[2]]:
                return False, f"si {r[0]} syllable times {r[2]} != {o[2]}"
            if len(r[3]) != len(o[3]):
                return False, f"si {r[0]} syllable count"
            for a, b in zip(r[3], o[3]):
                ok = same_f64(jsvec(a), b) if exact else rel_err(b, jsvec(a)) <= tol
                if not ok:
                    return False, f"si {r[0]} syllable features differ"
        elif level == 11:
            if not same_f64(jsvec(r[2]), o[2]):
                return False, f"callback time {r[2]} != {list(o[2])}"
            ok = same_f64(jsvec(r[3]), o[3]) if exact else rel_err(o[3], jsvec(r[3])) <= tol
            if not ok:
                return False, "utterance features differ"
        elif level == 10:
            if [list(x) for x in r[2]] != [list(x) for x in o[2]]:
                return False, f"si {r[0]} syllable times {r[2]} != {o[2]}"
            if len(r[3]) != len(o[3]):
                return False, f"si {r[0]} syllable count"
            for a, b in zip(r[3], o[3]):
                ra = np.array([[jsnum(x) for x in row] for row in a], dtype=np.float64).reshape(-1, 9)
                if not same_f64(ra, np.asarray(b, dtype=np.float64).reshape(-1, 9)):
                    return False, f"si {r[0]} syllable formant frames differ"
        elif level == 4:
            if not same_f64(jsvec(r[2]), o[2]):
                return False, f"si {r[0]} time"
            a = np.array([[jsnum(x) for x in row] for row in r[3]], dtype=np.float64)
            if not same_f64(a, np.asarray(o[3], dtype=np.float64)):
                return False, f"si {r[0]} formant frames differ"
    return True, ""
