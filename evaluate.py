#!/usr/bin/env python
"""Entry point kept from the reference (evaluate.py:65-119): same setup as train.py, then RunnerBase.evaluate(skip_reload=True)."""
import train

if __name__ == "__main__":
    train.main(evaluate=True)
