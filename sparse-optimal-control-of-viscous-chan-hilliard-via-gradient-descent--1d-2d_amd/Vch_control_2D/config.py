"""Parameter contract of the 2D solver: same field names, defaults and validation rules as
the reference's pydantic models (src/2D/Vch_control_2D/config.py:83-190), JSON save/load with
the same schema and default file name.  The interactive prompt editor (config.py:194-257) is
UI and is not part of the hot path; `get_user_input_for_config` is a non-interactive stand-in
that returns the previous instance (or defaults).
"""
from __future__ import annotations

import json
from typing import Optional, Type

from pydantic import BaseModel, Field, ValidationError, model_validator


class ForwardSolverConfig(BaseModel):
    """Grid, horizon and physical parameters (reference config.py:103-113)."""
    Nx: int = Field(128, gt=10)
    Ny: int = Field(128, gt=10)
    Lx: float = Field(1.0, gt=0)
    Ly: float = Field(1.0, gt=0)
    T: float = Field(1.0, gt=0)
    dt_initial: float = Field(1e-2, gt=0)
    tau: float = 0.05
    gamma: float = Field(10.0, gt=0)
    c1: float = 0.75
    c2: float = 1.0
    kappa: float = Field(0.01 ** 2, ge=0)

    @model_validator(mode="after")
    def _c2_above_c1(self):                      # reference config.py:115-120
        if self.c2 <= self.c1:
            raise ValueError(f"c2 ({self.c2}) must be greater than c1 ({self.c1})")
        return self


class OptimizationConfig(BaseModel):
    """Cost weights, step bound, iteration cap and control box (reference config.py:137-150)."""
    b1: float = Field(5.0, ge=0)
    b2: float = Field(10.0, ge=0)
    b3: float = Field(0.0001, ge=0)
    kappa_sparsity: float = Field(1e-4, ge=0)
    alpha_max: float = Field(50.0, gt=0)
    max_iter: int = Field(500, gt=10)
    u_min: float = -1.0
    u_max: float = 1.0

    @model_validator(mode="after")
    def _box(self):                              # reference config.py:146-150
        if self.u_max <= self.u_min:
            raise ValueError("u_max must be strictly greater than u_min.")
        return self


class SimulationParameters(BaseModel):
    """Container written to / read from JSON (reference config.py:153-157)."""
    forward_solver: ForwardSolverConfig = Field(default_factory=ForwardSolverConfig)
    optimization: OptimizationConfig = Field(default_factory=OptimizationConfig)
    last_run_iterations: int = 0


def save_params(fwd_config: ForwardSolverConfig, opt_config: OptimizationConfig, iteration_count: int,
                filepath: str = "last_run_config_2d.json") -> None:
    """Reference config.py:161-178."""
    params = SimulationParameters(forward_solver=fwd_config, optimization=opt_config,
                                  last_run_iterations=iteration_count)
    try:
        with open(filepath, "w") as f:
            f.write(params.model_dump_json(indent=4))
    except IOError as e:
        print(f"[Warning] Could not save configuration file: {e}")


def load_params(filepath: str = "last_run_config_2d.json") -> SimulationParameters:
    """Reference config.py:181-190: defaults when the file is missing or invalid."""
    try:
        with open(filepath, "r") as f:
            return SimulationParameters(**json.load(f))
    except (FileNotFoundError, ValidationError, json.JSONDecodeError):
        return SimulationParameters()


def get_user_input_for_config(config_model: Type[BaseModel], title: str = "",
                              previous_instance: Optional[BaseModel] = None) -> BaseModel:
    """Non-interactive stand-in for the prompt editor: previous values, else defaults."""
    return previous_instance if previous_instance is not None else config_model()


def get_yes_no_input(prompt: str) -> bool:
    """Batch mode: every confirmation is 'no' (keep parameters) — the UI is out of scope."""
    return False
