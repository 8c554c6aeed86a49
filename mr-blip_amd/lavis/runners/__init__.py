from lavis.runners.runner_base import RunnerBase

__all__ = ["RunnerBase"]
