#!/usr/bin/env python3
"""Generate tests/golden/pcg_solve.npz by RUNNING THE REFERENCE'S OWN ``solve()``.

`toast.ops.mapmaker_solve` cannot be imported here (it pulls in the compiled _libtoast), but its module-level
function ``solve`` (src/toast/ops/mapmaker_solve.py:524-755: the preconditioned conjugate gradient of the map-maker
with its convergence, stall and iteration-limit exits) is plain Python over the AmplitudesMap arithmetic.  This script
parses the reference file where it lies, compiles ONLY that one function definition from its syntax tree (decorator
dropped, nothing copied into the repository) and runs it on small dense symmetric positive definite systems:

    lhs_op.apply:        out = A @ in          (a dense matrix in place of M^T N^-1 Z M)
    apply_precond:       out = diag(A)^-1 in

The objects handed to it are minimal containers with the arithmetic the function calls (``-=``, ``+=``, ``*=``,
``dot``, ``duplicate``, ``reset``, ``.local``): NumPy expressions with the reference's order of operations
(amplitudes.py: a scaled temporary is added, dot = np.dot of the unflagged entries).  Every ``dot`` result is recorded,
which gives the exact sequence of residual norms (the function itself only logs them with seven digits).

Build container only; the fixture is committed.      python tests/golden/make_golden_pcg.py
"""
import ast
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/toast/ops/mapmaker_solve.py"


class Amp:
    """One template's amplitudes: the members solve() touches (reference: src/toast/templates/amplitudes.py)."""

    def __init__(self, n):
        self.local = np.zeros(n)
        self.n_global = n
        self.n_local = n

    def duplicate(self):
        out = Amp(self.n_local)
        out.local[:] = self.local
        return out

    def reset(self):
        self.local[:] = 0.0

    def clear(self):
        pass


class AmpMap(dict):
    dots = None     # list that records every dot product, in call order

    def duplicate(self):
        out = AmpMap()
        for k, v in self.items():
            out[k] = v.duplicate()
        return out

    def reset(self):
        for v in self.values():
            v.reset()

    def clear(self):
        super().clear()

    def __isub__(self, other):
        for k, v in self.items():
            v.local -= other[k].local
        return self

    def __iadd__(self, other):
        for k, v in self.items():
            v.local += other[k].local
        return self

    def __imul__(self, scalar):
        for v in self.values():
            v.local *= scalar
        return self

    def dot(self, other):
        val = 0.0
        for k, v in self.items():
            val += float(np.dot(v.local, other[k].local))
        AmpMap.dots.append(val)
        return val


class Data(dict):
    class _Comm:
        comm_world = None
        world_rank = 0

    comm = _Comm()


class DenseLHS:
    name = "dense"

    def __init__(self, A):
        self.A = A
        self.out = None
        outer = self

        class TM:
            amplitudes = None

            def apply_precond(self, amps_in, amps_out):
                amps_out["t"].local[:] = amps_in["t"].local / np.diag(outer.A)

        self.template_matrix = TM()

    def apply(self, data, detectors=None):
        data[self.out]["t"].local[:] = self.A @ data[self.template_matrix.amplitudes]["t"].local


class _Quiet:
    @staticmethod
    def get():
        return _Quiet()

    def __getattr__(self, name):
        return lambda *a, **k: None


def load_reference_solve():
    tree = ast.parse(open(REF).read(), REF)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "solve"]
    assert len(fn) == 1
    fn[0].decorator_list = []          # @function_timer
    mod = ast.Module(body=fn, type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = {"np": np, "Logger": _Quiet, "Timer": _Quiet, "AmplitudesMap": AmpMap}
    exec(compile(mod, REF, "exec"), ns)
    return ns["solve"]


def spd(rng, n, cond):
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    ev = np.geomspace(1.0, cond, n)
    a = (q * ev) @ q.T
    return 0.5 * (a + a.T)


def main():
    solve = load_reference_solve()
    rng = np.random.default_rng(20261002)
    out = {}
    cases = {
        # name: (n, condition number, convergence, n_iter_max, n_iter_min, starting guess?)
        "converges": (48, 1.0e3, 1.0e-16, 200, 3, False),
        "iteration_limit": (64, 1.0e6, 1.0e-30, 9, 3, False),
        "stalls": (40, 1.0e14, 1.0e-30, 300, 3, False),
        "starting_guess": (32, 1.0e2, 1.0e-20, 100, 3, True),
    }
    for name, (n, cond, conv, it_max, it_min, guess) in cases.items():
        A = spd(rng, n, cond)
        x_true = rng.standard_normal(n)
        b = A @ x_true
        data = Data()
        rhs = AmpMap(t=Amp(n))
        rhs["t"].local[:] = b
        data["rhs"] = rhs
        x0 = np.zeros(n)
        if guess:
            x0 = x_true + 0.1 * rng.standard_normal(n)
            start = AmpMap(t=Amp(n))
            start["t"].local[:] = x0
            data["result"] = start
        AmpMap.dots = []
        solve(data, None, DenseLHS(A), "rhs", "result", convergence=conv, n_iter_max=it_max, n_iter_min=it_min)
        dots = AmpMap.dots
        # call order: rhs.rhs, proposal.residual, then per iteration p.Ap, r.r [, z.r unless the loop ended]
        sq_init = dots[0]
        rr = dots[3::3]
        history = np.array(rr) / sq_init
        out.update({f"{name}_A": A, f"{name}_b": b, f"{name}_x0": x0, f"{name}_convergence": np.array(conv),
                    f"{name}_n_iter_max": np.array(it_max), f"{name}_n_iter_min": np.array(it_min),
                    f"{name}_history": history, f"{name}_solution": data["result"]["t"].local.copy(),
                    f"{name}_dots": np.array(dots)})
        print(f"{name:16s} n = {n:3d}  iterations {len(history):3d}  first {history[0]:.3e}  last {history[-1]:.3e}  "
              f"error {np.max(np.abs(data['result']['t'].local - x_true)):.2e}")
    np.savez_compressed(os.path.join(HERE, "pcg_solve.npz"), **out)


if __name__ == "__main__":
    main()
