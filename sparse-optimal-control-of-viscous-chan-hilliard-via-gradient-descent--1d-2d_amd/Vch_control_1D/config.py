"""Parameter contract of the 1D solver: same field names, defaults and validation rules as
src/1D/Vch_control_1D/config.py:91-171 (`N` instead of `Nx, Ny`), JSON save/load with the same
schema and default file name.  The interactive prompt editor is UI (out of scope);
`get_user_input_for_config` returns the previous instance or defaults."""
from __future__ import annotations

import json
from typing import Optional, Type

from pydantic import BaseModel, Field, ValidationError, model_validator


class ForwardSolverConfig(BaseModel):
    """Reference config.py:93-102 (1D)."""
    N: int = Field(128, gt=10)
    Lx: float = Field(1.0, gt=0)
    T: float = Field(1.0, gt=0)
    dt_initial: float = Field(1e-2, gt=0)
    tau: float = 0.05
    gamma: float = Field(10.0, gt=0)
    c1: float = 0.75
    c2: float = 1.0
    kappa: float = Field(0.03 ** 2, ge=0)

    @model_validator(mode="after")
    def _c2_above_c1(self):
        if self.c2 <= self.c1:
            raise ValueError(f"c2 ({self.c2}) must be greater than c1 ({self.c1})")
        return self


class OptimizationConfig(BaseModel):
    """Reference config.py:115-123 (1D)."""
    b1: float = Field(0.3, ge=0)
    b2: float = Field(13.0, ge=0)
    b3: float = Field(0.0019, ge=0)
    kappa_sparsity: float = Field(0.00009, ge=0)
    alpha_max: float = Field(100.0, gt=0)
    max_iter: int = Field(1000, gt=10)
    u_min: float = -1.0
    u_max: float = 1.0

    @model_validator(mode="after")
    def _box(self):
        if self.u_max <= self.u_min:
            raise ValueError("u_max must be strictly greater than u_min.")
        return self


class SimulationParameters(BaseModel):
    forward_solver: ForwardSolverConfig = Field(default_factory=ForwardSolverConfig)
    optimization: OptimizationConfig = Field(default_factory=OptimizationConfig)
    last_run_iterations: int = 0


def save_params(fwd_config, opt_config, iteration_count: int, filepath: str = "last_run_config.json") -> None:
    params = SimulationParameters(forward_solver=fwd_config, optimization=opt_config,
                                  last_run_iterations=iteration_count)
    try:
        with open(filepath, "w") as f:
            f.write(params.model_dump_json(indent=4))
    except IOError as e:
        print(f"[Warning] Could not save configuration file: {e}")


def load_params(filepath: str = "last_run_config.json") -> SimulationParameters:
    try:
        with open(filepath, "r") as f:
            return SimulationParameters(**json.load(f))
    except (FileNotFoundError, ValidationError, json.JSONDecodeError):
        return SimulationParameters()


def get_user_input_for_config(config_model: Type[BaseModel], title: str = "",
                              previous_instance: Optional[BaseModel] = None) -> BaseModel:
    return previous_instance if previous_instance is not None else config_model()


def get_yes_no_input(prompt: str) -> bool:
    return False
