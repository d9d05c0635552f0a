"""`mctf synthesize_step` entry point -- see qsvc_amd/cli/drivers.py."""
from .drivers import synthesize_step as main

if __name__ == "__main__":
    main()
