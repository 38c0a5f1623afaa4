"""Logging helpers with the reference's names (helmholtz_x/solver_utils.py:4-17)."""
import datetime
import os

_QUIET = os.environ.get("HX_QUIET", "0") == "1"


def start_time():
    return datetime.datetime.now()


def execution_time(start):
    print("Total Execution Time: ", datetime.datetime.now() - start)


def info(msg):
    if not _QUIET:
        print(msg)
