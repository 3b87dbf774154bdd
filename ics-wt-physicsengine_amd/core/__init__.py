"""GPU-backed stand-in for the hot path of ``wt_simulator.core``
(export list modelled on /root/reference/src/wt_simulator/core/__init__.py:207-263,
restricted to the multi-zone CSTR step and its batched pH solver)."""
from .reactor import (BoundaryConditions, EnsembleState, IntegratedCSTR, PhysicsEngine,
                      ReactorConfiguration, ReactorEnsemble, ReactorState, boundary_block)
from .chemistry import AqueousChemistry, BufferSystem, solve_pH
from .physics import (ArrheniusParameters, FlowParameters, GeometryParameters, SpatialModel, StratificationParameters,
                      TemperatureDependentKinetics, TransportModel, run_all_validations, validate_chemistry,
                      validate_integrated_reactor, validate_spatial, validate_thermodynamics, validate_transport)
from .synthetic import make_ensemble
from . import params, sharding
from .sharding import gather_state, shard_bounds

__all__ = ["BoundaryConditions", "EnsembleState", "IntegratedCSTR", "PhysicsEngine", "ReactorConfiguration",
           "ReactorEnsemble", "ReactorState", "boundary_block", "AqueousChemistry", "BufferSystem",
           "solve_pH", "make_ensemble", "params", "sharding", "gather_state", "shard_bounds",
           # the rest of wt_simulator.core's export list (core/__init__.py:238-263)
           "TemperatureDependentKinetics", "ArrheniusParameters", "TransportModel", "GeometryParameters", "FlowParameters",
           "SpatialModel", "StratificationParameters", "validate_thermodynamics", "validate_chemistry", "validate_transport",
           "validate_spatial", "validate_integrated_reactor", "run_all_validations"]
