"""`mctf texture_expand` entry point -- see qsvc_amd/cli/expand_tools.py."""
from .expand_tools import texture_expand as main

if __name__ == "__main__":
    main()
