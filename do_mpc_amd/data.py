"""`do_mpc.data`: storing and loading the per-step records of controller / simulator / estimator objects.

Reference: /root/reference/do_mpc/data.py:376-460 (`save_results`, `load_results`; the last lines of every examples/*/main.py
and of the reference's tests).  The records themselves are `do_mpc_amd.controller.MPCData` (one class for all three here)."""
import os
import pickle

from .controller import MPC, MPCData  # noqa: F401
from .simulator import Simulator


def save_results(save_list: list, result_name: str = "results", result_path: str = "./results/", overwrite: bool = False) -> None:
    """{'mpc': ..., 'simulator': ..., 'estimator': ...} -> `<result_path><result_name>.pkl`; without `overwrite` a taken name
    gets a running prefix `001_`, `002_`, ... like in the reference."""
    assert isinstance(save_list, list), "save_list must be a list."
    assert isinstance(result_name, str), "result_name must be a string."
    assert isinstance(result_path, str), "results_path must be a string."
    assert isinstance(overwrite, bool), "overwrite must be boolean."
    os.makedirs(result_path, exist_ok=True)
    results = {}
    for obj in save_list:
        if isinstance(obj, MPC):
            results["mpc"] = obj.data
        elif isinstance(obj, Simulator):
            results["simulator"] = obj.data
        elif hasattr(obj, "data") and type(obj).__name__ in ("StateFeedback", "EKF", "MHE"):
            results["estimator"] = obj.data
        else:
            raise Exception("save_list contains object which is neither do_mpc simulator, optimizizer nor estimator.")
    name = result_name
    if not overwrite:
        ind = 1
        while os.path.isfile(result_path + name + ".pkl"):
            name = "{ind:03d}_{name}".format(ind=ind, name=result_name)
            ind += 1
    with open(result_path + name + ".pkl", "wb") as f:
        pickle.dump(results, f)


def load_results(file_name: str) -> dict:
    with open(file_name, "rb") as f:
        return pickle.load(f)
