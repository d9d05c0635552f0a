"""`mctf synthesize` entry point -- see qsvc_amd/cli/drivers.py."""
from .drivers import synthesize as main

if __name__ == "__main__":
    main()
