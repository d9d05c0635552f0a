"""`mctf motion_expand_j2k` entry point -- see qsvc_amd/cli/expand_tools.py."""
from .expand_tools import motion_expand_j2k as main

if __name__ == "__main__":
    main()
