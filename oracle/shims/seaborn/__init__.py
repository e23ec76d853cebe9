"""Empty stand-in: /root/reference/utils.py:5 imports seaborn for plot helpers only."""
