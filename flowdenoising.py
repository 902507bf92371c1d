#!/usr/bin/env python3
"""flowdenoising.py -- drop-in command line (see flowdenoising_amd/cli.py for the option list)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from flowdenoising_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
