Fire = lambda f: None
