import os
import sys

from .osutils import mkdir_if_missing


class Logger(object):
    """Tee of stdout into a log file (`sys.stdout = Logger(path)` in the reference's scripts)."""

    def __init__(self, fpath=None):
        self.console = sys.stdout
        self.file = None
        if fpath is not None:
            mkdir_if_missing(os.path.dirname(fpath))
            self.file = open(fpath, 'w')

    def __del__(self):
        self.close()

    def __enter__(self):
        return self

    def __exit__(self, *args):
        self.close()

    def write(self, msg):
        self.console.write(msg)
        if self.file is not None:
            self.file.write(msg)

    def flush(self):
        self.console.flush()
        if self.file is not None:
            self.file.flush()
            os.fsync(self.file.fileno())

    def close(self):
        # the console stream is left open on purpose (it is the process's real stdout)
        if self.file is not None:
            self.file.close()
            self.file = None
