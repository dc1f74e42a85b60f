"""do_mpc_amd - MI355X-native structured interior-point backend behind do-mpc's MPC surface.

Namespaces mirror the reference package (`do_mpc.model.Model`, `do_mpc.controller.MPC`).
"""
from . import controller, differentiator, model, simulator, structs, sym  # noqa: F401
from .controller import MPC, MPCSettings  # noqa: F401
from .model import Model  # noqa: F401
from .simulator import Simulator  # noqa: F401

__version__ = "0.1.0"
