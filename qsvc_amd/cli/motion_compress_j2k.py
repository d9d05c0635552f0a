"""`mctf motion_compress_j2k` entry point -- see qsvc_amd/cli/motion_tools.py."""
from .motion_tools import motion_compress_j2k as main

if __name__ == "__main__":
    main()
