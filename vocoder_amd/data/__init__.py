"""Mirror of the one piece of fish_vocoder/data that sits directly before the generator path: the log-mel front-end."""
