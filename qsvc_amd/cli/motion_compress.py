"""`mctf motion_compress` entry point -- see qsvc_amd/cli/motion_tools.py."""
from .motion_tools import motion_compress as main

if __name__ == "__main__":
    main()
