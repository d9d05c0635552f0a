"""`mctf bidirectional_motion_decorrelate` entry point -- see qsvc_amd/cli/motion_tools.py."""
from .motion_tools import bidirectional_motion_decorrelate as main

if __name__ == "__main__":
    main()
