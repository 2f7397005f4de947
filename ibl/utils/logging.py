"""`sys.stdout = Logger(path)` (examples/test.py:89): a tee — everything printed goes to the stream
that was sys.stdout at construction time and, when a path is given, to that file as well."""
import os
import sys


class Logger(object):
    def __init__(self, fpath=None):
        self._sinks = [sys.stdout]
        if fpath is not None:
            folder = os.path.dirname(fpath)
            if folder:
                os.makedirs(folder, exist_ok=True)
            self._sinks.append(open(fpath, 'w'))

    # the two attributes callers of the reference's class may look at
    console = property(lambda self: self._sinks[0])
    file = property(lambda self: self._sinks[1] if len(self._sinks) > 1 else None)

    def write(self, msg):
        for s in self._sinks:
            s.write(msg)
        return len(msg)

    def flush(self):
        for s in self._sinks:
            s.flush()

    def close(self):
        """Closes the log file only: the first sink is the process's real stdout."""
        while len(self._sinks) > 1:
            self._sinks.pop().close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __getattr__(self, name):
        # isatty / encoding / fileno ...: whatever else is asked of a stdout replacement
        return getattr(self._sinks[0], name)
