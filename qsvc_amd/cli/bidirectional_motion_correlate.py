"""`mctf bidirectional_motion_correlate` entry point -- see qsvc_amd/cli/expand_tools.py."""
from .expand_tools import bidirectional_motion_correlate as main

if __name__ == "__main__":
    main()
