"""Sync-directory naming of the file-based master/worker transport.

The wire protocol (reference agents/GTN_base.py:13-29) is a set of file names under `<cwd>/results/GTN_sync`:
`<bohb_id>_<id>_{input,input_check,result,result_check}.pt` plus `quit.flag`.  Only those strings are shared with the
reference; masters and workers of either implementation can therefore meet in the same directory."""
import os

_SYNC_SUBDIR = os.path.join("results", "GTN_sync")
_KINDS = ("input", "input_check", "result", "result_check")


class GTN_Base(object):
    def __init__(self, bohb_id):
        self.bohb_id = bohb_id
        self.sync_dir = os.path.join(os.getcwd(), _SYNC_SUBDIR)
        os.makedirs(self.sync_dir, exist_ok=True)

    def _sync_file(self, id, kind):
        assert kind in _KINDS
        return os.path.join(self.sync_dir, "%s_%s_%s.pt" % (self.bohb_id, id, kind))

    def get_input_file_name(self, id):
        return self._sync_file(id, "input")

    def get_input_check_file_name(self, id):
        return self._sync_file(id, "input_check")

    def get_result_file_name(self, id):
        return self._sync_file(id, "result")

    def get_result_check_file_name(self, id):
        return self._sync_file(id, "result_check")

    def get_quit_file_name(self):
        return os.path.join(self.sync_dir, "quit.flag")

    def worker_files(self, id):
        """All four transport files of worker `id`."""
        return [self._sync_file(id, k) for k in _KINDS]

    def clean_working_dir(self):
        with os.scandir(self.sync_dir) as entries:
            for entry in entries:
                if entry.is_file():
                    os.remove(entry.path)
