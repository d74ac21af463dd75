"""Engine-context cache for the function-style mirror modules: the reference's functions are
stateless, the GPU engine is not, so contexts are keyed by (grid, parameters, batch)."""
from __future__ import annotations

from ..engine import Engine2D

_CACHE = {}
_MAX = 4


def engine_for(Nx, Ny, Lx, Ly, tau, gamma, c1, c2, kappa, batch=1, max_steps=128, device=0) -> Engine2D:
    key = (int(Nx), int(Ny), float(Lx), float(Ly), float(tau), float(gamma), float(c1), float(c2),
           float(kappa), int(batch), int(device))
    e = _CACHE.get(key)
    if e is not None and e.max_steps >= max_steps:
        return e
    if e is not None:
        e.close()
        del _CACHE[key]
    while len(_CACHE) >= _MAX:                     # histories are GBs: keep few contexts alive
        k0 = next(iter(_CACHE))
        _CACHE.pop(k0).close()
    e = Engine2D(Nx, Ny, Lx, Ly, tau, gamma, c1, c2, kappa, batch=batch, max_steps=max_steps, device=device)
    _CACHE[key] = e
    return e


def engine_for_config(cfg, batch=1, max_steps=128, device=0) -> Engine2D:
    return engine_for(cfg.Nx, cfg.Ny, cfg.Lx, cfg.Ly, cfg.tau, cfg.gamma, cfg.c1, cfg.c2, cfg.kappa,
                      batch=batch, max_steps=max_steps, device=device)


def clear():
    for e in _CACHE.values():
        e.close()
    _CACHE.clear()
