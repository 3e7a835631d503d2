"""Console prefixes used by the entry point (reference: tools/common.py:15-51)."""
import time


class _Notify(object):
    ENDC = '\033[0m'

    def _stamp(self, colour, tag):
        return '%s[%s %s]' % (colour, tag, time.strftime('%H:%M:%S'))

    @property
    def INFO(self):
        return self._stamp('\033[94m', 'INFO')

    @property
    def WARNING(self):
        return self._stamp('\033[93m', 'WARNING')

    @property
    def FAIL(self):
        return self._stamp('\033[91m', 'ERROR')


Notify = _Notify()
