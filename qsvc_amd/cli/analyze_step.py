"""`mctf analyze_step` entry point -- see qsvc_amd/cli/drivers.py."""
from .drivers import analyze_step as main

if __name__ == "__main__":
    main()
