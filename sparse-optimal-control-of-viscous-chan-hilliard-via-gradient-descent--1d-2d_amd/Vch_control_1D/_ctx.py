"""Engine-context cache for the 1D function-style mirror modules."""
from __future__ import annotations

from ..engine import Engine1D

_CACHE = {}


def engine_for(N, Lx=1.0, tau=0.05, gamma=10.0, c1=0.75, c2=1.0, kappa=0.03 ** 2, batch=1, max_steps=128,
               device=0) -> Engine1D:
    key = (int(N), float(Lx), float(tau), float(gamma), float(c1), float(c2), float(kappa), int(batch), int(device))
    e = _CACHE.get(key)
    if e is not None and e.max_steps >= max_steps:
        return e
    if e is not None:
        e.close()
    while len(_CACHE) >= 6:
        _CACHE.pop(next(iter(_CACHE))).close()
    e = Engine1D(N, Lx, tau, gamma, c1, c2, kappa, batch=batch, max_steps=max_steps, device=device)
    _CACHE[key] = e
    return e


def engine_for_config(cfg, batch=1, max_steps=128, device=0) -> Engine1D:
    return engine_for(cfg.N, cfg.Lx, cfg.tau, cfg.gamma, cfg.c1, cfg.c2, cfg.kappa, batch=batch,
                      max_steps=max_steps, device=device)


def clear():
    for e in _CACHE.values():
        e.close()
    _CACHE.clear()
