"""`mctf compress` entry point -- see qsvc_amd/cli/drivers.py."""
from .drivers import compress as main

if __name__ == "__main__":
    main()
