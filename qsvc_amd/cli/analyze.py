"""`mctf analyze` entry point -- see qsvc_amd/cli/drivers.py."""
from .drivers import analyze as main

if __name__ == "__main__":
    main()
